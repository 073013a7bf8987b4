cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "bf16_mode_against or fp8_weight_only or sampler_golden" 2>&1 | grep -E "passed|failed|d0|C5|Error|assert" | head -20
