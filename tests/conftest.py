import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: dump the observed error / stated tolerance of every parity comparison (max per call site)."""
    try:
        import problems as P
    except Exception:
        return
    if not P.PARITY_LOG:
        return
    import json
    worst = {}
    for rec in P.PARITY_LOG:
        key = f"{rec.get('file', '')}:{rec['test']}:{rec['line']}:{rec['dtype']}:{rec.get('form', '')}" + (f":{rec['what']}" if rec["line"] == 0 else "")
        if key not in worst or rec["ratio"] > worst[key]["ratio"]:
            worst[key] = rec
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    keyof = lambda rec: f"{rec.get('file', '')}:{rec['test']}:{rec['line']}:{rec['dtype']}:{rec.get('form', '')}" + (f":{rec['what']}" if rec["line"] == 0 else "")
    if os.environ.get("CIAO_PARITY_LOG_MERGE") == "1" and os.path.exists(os.path.join(out, "parity_observed.json")):
        # several sessions into one log (the seeded random tests under several seeds: tools/exp/calibrate_seeds.sh): the worst per call site
        for rec in json.load(open(os.path.join(out, "parity_observed.json"))):
            k = keyof(rec)
            if k not in worst or rec["ratio"] > worst[k]["ratio"]:
                worst[k] = rec
    with open(os.path.join(out, "parity_observed.json"), "w") as fh:
        json.dump(sorted(worst.values(), key=lambda r: (r["test"], r["line"], r["dtype"])), fh, indent=1)


@pytest.fixture(scope="session")
def ciao():
    """The product package, loaded under its importable alias."""
    import ciao_loader
    return ciao_loader.load()


@pytest.fixture(scope="session")
def ctx(ciao):
    """One library context on cuda:0 for the whole GPU session (fails loudly without the HIP extension / a GPU)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ciaoalgorithms_jl_amd.device import Context
    c = Context(0)
    for kv in os.environ.get("CIAO_TEST_OPTS", "").split(","):      # tuning experiments: run the suite with a library option set
        if "=" in kv:
            k, v = kv.split("=")
            c.set_option(k, int(v))
    yield c
    c.synchronize()
    c.close()
