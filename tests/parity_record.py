"""Achieved parity errors of the config-size GPU tests, written to profiles/r6_parity.json (round 4: r4_parity.json) (the
whole-iteration tests printed their errors and threw them away).  One entry per test name: max abs errors per tensor,
sizes, seed.  The file is merged, not overwritten, so one pytest run (or several gpurun calls) accumulate into it; on
the GPU box it is written under gpurun_out/ as well so that it travels back."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATHS = [os.path.join(ROOT, "profiles", "r6_parity.json"), os.path.join(ROOT, "gpurun_out", "r6_parity.json")]


def record(name, errors, sizes=None, seed=None, note=None):
    def num(v):
        return v if isinstance(v, (list, str)) or v is None else float(v)
    entry = {"errors": {k: num(v) for k, v in errors.items()}, "sizes": sizes, "seed": seed}
    if note:
        entry["note"] = note
    for path in PATHS:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            data = {}
            if os.path.exists(path):
                with open(path) as f:
                    data = json.load(f)
            data[name] = entry
            with open(path, "w") as f:
                json.dump(data, f, indent=1, sort_keys=True)
        except OSError:
            pass
    return entry
