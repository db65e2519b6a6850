// gfx950 global_load_lds_dwordx4: where does lane L's 16 bytes land?  Expectation checked here: LDS address =
// wave-uniform base (M0) + 16 * lane, i.e. a wave copies 1 KB of global memory to 1 KB of LDS verbatim.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_direct_load.hip -o /tmp/ldl && /tmp/ldl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void copy_kernel(const char* __restrict__ src, unsigned* out, int perm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gptr;
    const int src_lane = perm ? (lane ^ 5) : lane;       // a lane may fetch any 16 bytes; they land at ITS slot
    __builtin_amdgcn_global_load_lds((gptr*)(src + wv * 1024 + src_lane * 16), (lds_void*)(smem + wv * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < blockDim.x * 4; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}

int main() {
    const int threads = 512, bytes = threads * 16;
    std::vector<unsigned> h(bytes / 4), o(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    char* d; unsigned* dout;
    hipMalloc(&d, bytes); hipMalloc(&dout, bytes);
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    for (int perm = 0; perm < 2; ++perm) {
        hipMemset(dout, 0, bytes);
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(threads), bytes, 0, d, dout, perm);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(o.data(), dout, bytes, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < threads / 64; ++w)
            for (int l = 0; l < 64; ++l)
                for (int k = 0; k < 4; ++k) {
                    const int sl = perm ? (l ^ 5) : l;
                    if (o[w * 256 + l * 4 + k] != h[w * 256 + sl * 4 + k]) ++bad;
                }
        printf("perm=%d: %s (%d mismatching dwords)\n", perm, bad ? "UNEXPECTED LAYOUT" : "lane L's 16 bytes land at base + 16*L", bad);
        if (bad) { for (int i = 0; i < 16; ++i) printf("%08x ", o[i]); printf("\n"); }
    }
    return 0;
}
