"""Compact print of scripts/wino_ab.py's per-shape JSON lines (stdin): one row per shape, `variant:ms` columns."""
import json
import sys

for line in sys.stdin:
    key, _, rest = line.partition(" ")
    try:
        d = json.loads(rest)
    except Exception:
        if "amdgpu.ids" not in line:
            print(line.strip())
        continue
    print(key, " ".join(f"{v}:{d[v]['ms']}{'' if d[v].get('bit_equal_to_first', True) else '(DIFF)'}" for v in d))
