timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -3
