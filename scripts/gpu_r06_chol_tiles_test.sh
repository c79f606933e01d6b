cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_gpu_chol_tiles.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r06_chol_tiles_test.log
