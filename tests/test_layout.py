"""CPU: repository contract checks -- the oracle is test infrastructure only; the product never
imports it; no reference sources or CUDA compatibility shims in the tree."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(root):
    for d, _, files in os.walk(root):
        if "__pycache__" in d:
            continue
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_imports_the_oracle():
    for f in _py_files(os.path.join(REPO, "cmflow_amd")):
        text = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
        assert "liboracle" not in text, f


def test_only_allowed_files_touch_the_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(REPO):
        if f.endswith(".py") and f not in allowed:
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(REPO, f)).read(), flags=re.M), f
    bench = open(os.path.join(REPO, "bench.py")).read()
    # in bench.py the oracle is imported only inside the cpu_baseline leg
    body = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    assert "from oracle" in body and "from oracle" not in bench.replace(body, "")


def test_oracle_headers_say_test_infrastructure():
    for f in ("cmf_oracle.c", "ops.py", "cmflow_oracle.py", "train_oracle.py", "__init__.py"):
        assert "TEST INFRASTRUCTURE ONLY" in open(os.path.join(REPO, "oracle", f)).read(), f


def test_no_compat_layers_in_kernels():
    for f in os.listdir(os.path.join(REPO, "cmflow_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(REPO, "cmflow_amd", "csrc", f)).read()
            assert "__HIP_PLATFORM" not in text and "cuda_runtime" not in text and "hipify" not in text.lower(), f


def test_nothing_reads_the_reference_at_run_time():
    """/root/reference does not exist on the GPU box: only the golden generator may name it."""
    for root in ("cmflow_amd", "oracle"):
        for f in _py_files(os.path.join(REPO, root)):
            assert "/root/reference" not in open(f).read(), f
    for f in ("bench.py", "__graft_entry__.py"):
        text = open(os.path.join(REPO, f)).read()
        assert "sys.path.insert(0, \"/root/reference\")" not in text and "open(\"/root/reference" not in text


def test_product_has_no_torch_math_path():
    """The block modules of the product run on the HIP kernels only: the torch-math twin of the point-major path is a test
    fixture (tests/pm_torch.py), not a dispatch option inside cmflow_amd/ (no vendor GEMM / BatchNorm behind the modules)."""
    for f in ("radarflow_util.py", "cmflow.py", "raflow.py", "fused.py", "fused_blocks.py"):
        text = open(os.path.join(REPO, "cmflow_amd", f)).read()
        assert "F.linear" not in text and "batch_norm" not in text and "use_blocks" not in text and '"pm_torch"' not in text, f


def test_product_has_one_loss_path():
    """cmflow_amd/losses.py runs the fused kernels only (cmf_radar_loss, cmf_pseudo_labels) and raises outside their range: the
    torch-op restatement of losses/radar_loss.py is a test fixture (tests/loss_torch.py), not a fallback inside the product."""
    text = open(os.path.join(REPO, "cmflow_amd", "losses.py")).read()
    for needle in ("torch.topk", "square_distance", "F.relu", "binary_cross_entropy", "torch.softmax", "native=", "index_points_group"):
        assert needle not in text, needle
    assert "raise RuntimeError" in text and "cmf_radar_loss" in text
    assert os.path.exists(os.path.join(REPO, "tests", "loss_torch.py"))
