cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "reference_keyed or bcast_weights" 2>&1 | grep -v -E "RCCL version|HIP version|ROCm version|Hostname|Librccl" | tail -25
