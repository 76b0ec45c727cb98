"""Parity figures measured by the -m gpu tests, written to gpurun_out/parity_record.json (scratch) so that a round's numbers can be
committed as profiles/rNN/parity.json; the tests assert against FIXED tolerances derived from that committed file (PARITY below)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_OUT = os.environ.get("RSU_PARITY_RECORD", os.path.join(ROOT, "gpurun_out", "parity_record.json"))


def record(name, **figures):
    try:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        data = json.load(open(_OUT)) if os.path.exists(_OUT) else {}
        data[name] = {k: (float(v) if not isinstance(v, (str, list, dict)) else v) for k, v in figures.items()}
        json.dump(data, open(_OUT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def committed():
    """the latest committed profiles/rNN/parity.json (measured figures the fixed tolerances below were derived from), or {}"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "parity.json")))
    return json.load(open(files[-1])) if files else {}
