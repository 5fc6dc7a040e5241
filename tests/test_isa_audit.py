"""ISA-level guards (CPU only: hipcc cross-compiles gfx950 without a GPU).  r03 found two kernels whose source read like "all
loads, then use them" while the compiler had put every load in its own branch with a full `s_waitcnt vmcnt(0)` behind it -
8 dependent memory round trips in the fused optimizer's id scan (3.2 us of a 13.7 us launch), 4 in the fused tower's input
tile.  scratch/audit_serial_loads.py counts such chains in the built ISA; these tests keep them out."""
import importlib.util
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
CSRC = ROOT / "two_tower_amazon_recommender_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _audit():
    spec = importlib.util.spec_from_file_location("audit_serial_loads", ROOT / "scratch" / "audit_serial_loads.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _isa(tmp_path, name):
    if not pathlib.Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    out = tmp_path / f"{name}.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT / 'include'}", "-S",
                    "--cuda-device-only", "-o", str(out), str(CSRC / f"{name}.hip")], check=True, capture_output=True, timeout=600)
    return out


def test_fused_optimizer_scan_and_routing_have_no_chain_of_single_dependent_loads(tmp_path):
    audit = _audit()
    c = audit.chains(_isa(tmp_path, "sparse"))
    opt = {k: v for k, v in c.items() if k.startswith("optimizer_ids_kernel")}
    assert len(opt) == 8
    for name, (_, longest) in opt.items():
        # what is left are the ranked / hot-range paths' pointer-chasing loops (<= 3); the id scan itself had 8
        assert longest <= 3, (name, longest)
    r = audit.chains(_isa(tmp_path, "route"))
    assert all(longest < 2 for k, (_, longest) in r.items() if k.startswith("route_kernel")), r


def test_fused_tower_input_tile_has_no_chain_of_single_dependent_loads(tmp_path):
    audit = _audit()
    c = audit.chains(_isa(tmp_path, "tower"))
    tw = {k: v for k, v in c.items() if k.startswith("tower_fwd2_kernel")}
    assert len(tw) == 8
    assert all(longest < 2 for _, longest in tw.values()), tw
