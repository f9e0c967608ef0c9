for g in 512 384 256 128; do echo "PP_WF_GRID=$g"; PP_WF_GRID=$g python tools/diag_wavefront.py 64 4096 2>&1 | grep standalone; done
