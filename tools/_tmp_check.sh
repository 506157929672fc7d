timeout 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "conv" --timeout 200 --timeout-method=thread 2>&1 | tail -2
for i in 1 2; do timeout 200 python tools/key_sections.py 2>/dev/null | grep -E "backbone|whole key|small net|whole non-key"; done
