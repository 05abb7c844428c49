"""CPU: the product never reaches into the test infrastructure (oracle/) and ships its own marker-weight table."""
import ast
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_imports(path):
    """(line, enclosing function) of every import of the oracle package in a source file."""
    tree = ast.parse(open(path).read(), path)
    hits = []

    def visit(node, fn):
        for ch in ast.iter_child_nodes(node):
            f = ch.name if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(ch, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in ch.names):
                hits.append((ch.lineno, fn))
            if isinstance(ch, ast.ImportFrom) and (ch.module or "").split(".")[0] == "oracle":
                hits.append((ch.lineno, fn))
            visit(ch, f)

    visit(tree, None)
    return hits


def test_product_does_not_import_the_oracle():
    files = [os.path.join(ROOT, f) for f in ("run.py", "run_inference.py", "miphei_vit_amd.py")]
    for d, _, names in os.walk(os.path.join(ROOT, "miphei-vit_amd")):
        files += [os.path.join(d, n) for n in names if n.endswith(".py")]
    for f in files:
        assert _oracle_imports(f) == [], f
        assert "oracle" not in open(f).read(), f      # not even through importlib / a string
    # bench.py: only the cpu_baseline leg (the checker) may use it
    hits = _oracle_imports(os.path.join(ROOT, "bench.py"))
    assert hits and all(fn in ("cpu_baseline", "cpu_baseline_tiny") for _, fn in hits), hits


def test_orion_marker_weights_come_from_the_shipped_stats_file():
    from miphei_vit_amd.config import compose
    from miphei_vit_amd.loss import marker_weights_from_file
    from oracle.model import orion_marker_weights
    cfg = compose(os.path.join(ROOT, "configs"), ["+default_configs=miphei-vit"])
    w = marker_weights_from_file(os.path.join(ROOT, cfg.data.channel_stats_path), cfg.data.targ_channel_names)
    assert w.shape == (16,) and float(w.min()) == 1.0
    assert torch.allclose(w, orion_marker_weights(16), atol=5e-5)     # SURVEY.md section 8(d), 4 decimals
