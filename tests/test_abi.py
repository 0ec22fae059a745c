"""CPU: the C-ABI library loads and exports every symbol include/megaverse_hip.h declares; the
product fails loudly without a HIP device; nothing in the product path touches oracle/."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "megaverse_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import megaverse_amd.extension as ext
    lib = ext.load_library()
    names = declared_symbols()
    assert len(names) >= 40
    bound = {n for n, _, _ in ext.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/megaverse_hip.h but not exported"
        assert n in bound, f"{n} missing from megaverse_amd.extension.SYMBOLS"
    assert bound <= set(names)


def test_action_space_sizes_no_device_needed():
    import megaverse_amd.extension as ext
    out = (C.c_int32 * 6)()
    assert ext.load_library().mv_action_space_sizes(out) == 0
    assert list(out) == [3, 3, 3, 2, 2, 3]     # env.cpp:33


def test_no_cpu_fallback():
    """on a box without a GPU construction must raise, never silently simulate on the host"""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    from megaverse_amd import MegaverseGym
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        MegaverseGym("TowerBuilding", 128, 72, 2, 1, 1, False, {})


def test_unknown_scenario_is_an_error_not_exit():
    # reference: Scenario::create -> TLOG(FATAL) -> exit(-1) (scenario.hpp:71-72); here: status code
    import megaverse_amd.extension as ext
    lib = ext.load_library()
    cfg = ext._Config(b"NoSuchScenario", 128, 72, 1, 1, 1, 0, 0, None, None, 0, 0, 0)
    h = C.c_void_p()
    assert lib.mv_create(C.byref(cfg), C.byref(h)) < 0
    assert b"Unknown scenario" in lib.mv_last_error()


def test_product_sources_do_not_use_the_oracle():
    """the product may not include, import, link or dlopen anything under oracle/ (comments may name it)"""
    pkg = os.path.join(ROOT, "megaverse_amd")
    bad = re.compile(r"#\s*include[^\n]*oracle|import\s+oracle|from\s+oracle|oracle_lib|libmv_oracle|mvo_[a-z]|-lmv_oracle|oracle/[a-z_]+\.(so|cpp|h)\b")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                m = bad.search(txt)
                assert m is None, f"{f}: {m.group(0)!r}"
    for f in ("bench.py",):      # bench.py may only use it inside its cpu_baseline*() legs
        txt = open(os.path.join(ROOT, f)).read()
        funcs = re.split(r"\n(?=def |class |if __name__)", txt)          # top-level blocks
        assert any(b.startswith("def cpu_baseline") for b in funcs)
        for b in funcs[1:]:
            if not b.startswith("def cpu_baseline"):
                assert "oracle" not in b, b[:60]


REFERENCE_TABLE = ["num_agents", "action_space_sizes", "seed", "reset", "set_actions", "step", "is_done", "get_observation",
                   "get_last_rewards", "true_objective", "set_render_resolution", "draw_hires", "draw_overview",
                   "get_hires_observation", "get_reward_shaping", "set_reward_shaping", "close"]   # bindings/megaverse.cpp:274-291


def test_pybind_module_has_the_reference_table():
    """the pybind11 flavour of the binding (megaverse_amd/pybind): same module surface as megaverse.extension.megaverse"""
    from megaverse_amd import build
    build.build_pybind()
    from megaverse_amd.pybind import megaverse as m
    assert callable(m.set_megaverse_log_level)
    for name in REFERENCE_TABLE:
        assert callable(getattr(m.MegaverseGym, name)), name
    from conftest import _has_gpu
    if not _has_gpu():   # no device: construction must fail loudly, never fall back to a host implementation
        with pytest.raises(RuntimeError, match="no HIP device"):
            m.MegaverseGym("TowerBuilding", 128, 72, 1, 1, 1, False, {})


def test_debug_snapshot_record_has_the_oracles_layout():
    """tests compare mv_debug_snapshot and mvo_snapshot byte for byte: the two packed records must have one size"""
    import oracle_lib
    from megaverse_amd import extension as ext
    assert ext.load_library().mv_debug_snapshot_size(None) == oracle_lib.SNAP.itemsize
