MPNN_EXTRA_HIPCC_FLAGS=-DMT_STAMP python -m mpnn_amd.build --force > /dev/null 2>&1
python tools/_stamp.py
