"""print the headline fields of bench.py JSON lines: python tools/show_bench.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        keys = ("value", "ms_per_step", "repeats", "ms_per_step_min", "ms_per_step_max", "wall_ms_per_step")
        print(f, {k: (round(d[k], 5) if isinstance(d.get(k), float) else d.get(k)) for k in keys})
        r = d.get("roofline", {})
        print("   roofline:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()
                               if k in ("achieved", "frac", "kernel_us", "step_frac", "grad_only_kernel_us", "traffic")})
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e, open(f).read()[-800:])
