"""CPU: the C-ABI library loads and exports every symbol include/sbr_hip.h declares; without a GPU
the engine refuses to run (no CPU fallback) instead of computing anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import LOSS_HINGE, hparams
from sbr_rs_amd import _lib
from sbr_rs_amd._abi import ModelKind, SbrHparams, Status

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_header():
    text = open(os.path.join(ROOT, "include", "sbr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from sbr_rs_amd import build

        build.build(verbose=False)
    L = _lib.load()
    declared = _declared_in_header()
    assert len(declared) >= 30
    assert sorted(_lib.DECLARED_SYMBOLS) == declared, "loader table and header disagree"
    for name in declared:
        assert hasattr(L, name), name
    from sbr_rs_amd._abi import ABI_VERSION

    assert L.sbr_abi_version() == ABI_VERSION
    assert b"No interactions" in L.sbr_status_string(int(Status.NO_INTERACTIONS))


def test_hparams_struct_layout_matches_header():
    assert C.sizeof(SbrHparams) == 68
    assert SbrHparams.seed.offset == 36 and SbrHparams.batch_sequences.offset == 64


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly():
    from sbr_rs_amd.engine import Model
    from sbr_rs_amd.errors import EngineError

    with pytest.raises(EngineError) as e:
        Model(hparams(10, 8, 16, int(ModelKind.EWMA), LOSS_HINGE))
    assert e.value.status == Status.NO_DEVICE
    x = np.zeros(4, np.float32)
    assert _lib.load().sbr_selftest_math(x.ctypes.data_as(C.c_void_p), 4, None, None, None) == Status.NO_DEVICE


def test_python_surface_mirrors_reference_names():
    import sbr_rs_amd as sbr

    h = sbr.lstm.Hyperparameters.new(100, 32)
    for setter in ("learning_rate", "l2_penalty", "embedding_dim", "num_epochs", "loss", "lstm_variant", "num_threads",
                   "parallelism", "rng", "from_seed", "optimizer", "build"):
        assert callable(getattr(h, setter)), setter
    assert callable(sbr.lstm.Hyperparameters.random) and callable(sbr.ewma.Hyperparameters.random)
    assert not hasattr(sbr.ewma.Hyperparameters.new(10, 8), "lstm_variant")
    for name in ("user_based_split", "train_test_split", "Interaction", "Interactions", "CompressedInteractions", "TripletInteractions"):
        assert hasattr(sbr.data, name)
    assert callable(sbr.evaluation.mrr_score)
    assert issubclass(sbr.FittingError.NoInteractions, sbr.FittingError)
    assert issubclass(sbr.PredictionError.InvalidPredictionValue, sbr.PredictionError)
    r = sbr.lstm.Hyperparameters.random(50, sbr.XorShiftRng.from_seed(bytes([1] * 16)))
    assert 16 <= r._item_embedding_dim <= 128 and 16 <= r._max_sequence_length <= 128


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 on its own (no C++, no HIP, no torch types)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('#include "sbr_hip.h"\n'
                   "int probe(void) { sbr_hparams hp; sbr_model* m = 0; (void)m; return (int)sizeof(hp); }\n"
                   "sbr_status (*const fit_ptr)(sbr_model*, const uint64_t*, const uint32_t*, uint64_t, float*) = sbr_model_fit;\n")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "abi.o")])


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/ (it is the
    checker); nothing under sbr_rs_amd/ or include/ may mention it, and nothing the GPU runs reads
    /root/reference."""
    offenders = []
    for base in ("sbr_rs_amd", "include"):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if not f.endswith((".py", ".hip", ".h", ".hpp")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "libsbr_oracle" in text or "orc_" in text:
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle", bench)]
    lo, hi = bench.index("def cpu_baseline"), bench.index("def movielens_mrr")
    assert uses and all(lo < u < hi for u in uses), "bench.py may use the oracle only inside cpu_baseline()"
    for f in ("bench.py", "__graft_entry__.py"):
        code = re.sub(r'""".*?"""', "", open(os.path.join(ROOT, f)).read(), flags=re.S)
        code = re.sub(r"#.*", "", code)
        assert "/root/reference" not in code, f


def test_oracle_shares_only_the_approximation_header():
    """The oracle states the scalar formulas itself (oracle/orc_numerics.h); from the product it may include
    nothing but the approximation polynomial (sbr_rs_amd/csrc/sbr_approx.h), and from include/ only the ABI
    header (types and enum values)."""
    odir = os.path.join(ROOT, "oracle")
    for f in os.listdir(odir):
        if not f.endswith((".c", ".h")):
            continue
        for inc in re.findall(r'#\s*include\s*"([^"]+)"', open(os.path.join(odir, f)).read()):
            target = os.path.normpath(os.path.join(odir, inc))
            rel = os.path.relpath(target, ROOT)
            if rel.startswith("sbr_rs_amd"):
                assert rel == os.path.join("sbr_rs_amd", "csrc", "sbr_approx.h"), (f, inc)
            elif rel.startswith("include"):
                assert rel == os.path.join("include", "sbr_hip.h"), (f, inc)
            else:
                assert rel.startswith("oracle"), (f, inc)
    approx = open(os.path.join(ROOT, "sbr_rs_amd", "csrc", "sbr_approx.h")).read()
    code = re.sub(r"/\*.*?\*/", "", approx, flags=re.S)
    # nothing but the polynomial lives there: no cell, loss, optimiser or generator
    for word in ("adagrad", "adam", "lstm", "hinge", "bpr", "xorshift", "sigmoid"):
        assert word not in code.lower(), word
