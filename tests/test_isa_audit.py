"""ISA-level guards (CPU only: hipcc cross-compiles gfx950 without a GPU; helpers in tests/isa_audit/).

* Serial load chains (r03): two kernels read like "all loads, then use them" in the source while the compiler had put every load in
  its own branch with a full `s_waitcnt vmcnt(0)` behind it - 8 dependent memory round trips in the fused optimizer's id scan
  (3.2 us of a 13.7 us launch), 4 in the fused tower's input tile.  audit_serial_loads.py counts such chains in the built ISA.
* Barrier loops (r02 note, r03 audit, a test since r04): every wave of a workgroup must reach each `s_barrier` the same number of
  times.  hipcc guarantees that only for control flow it KNOWS to be wave-uniform, which shows in the ISA as scalar loop control.
  audit_barriers.py checks every score_kernel instantiation: each loop around an `s_barrier` closes and exits on scalar branches
  and no barrier can be jumped over under a lane mask; its negative control (a barrier loop on a per-lane trip count) must be flagged.
"""
import importlib.util
import pathlib
import re
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
CSRC = ROOT / "two_tower_amazon_recommender_amd" / "csrc"
AUDIT = ROOT / "tests" / "isa_audit"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _mod(name):
    spec = importlib.util.spec_from_file_location(name, AUDIT / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _isa(tmp_path, src: pathlib.Path, extra=()):
    if not pathlib.Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    out = tmp_path / f"{src.stem}{'_'.join(extra).replace('-', '').replace('=', '')}.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT / 'include'}", *extra, "-S",
                    "--cuda-device-only", "-o", str(out), str(src)], check=True, capture_output=True, timeout=900)
    return out


def test_fused_optimizer_scan_and_routing_have_no_chain_of_single_dependent_loads(tmp_path):
    audit = _mod("audit_serial_loads")
    c = audit.chains(_isa(tmp_path, CSRC / "sparse.hip"))
    opt = {k: v for k, v in c.items() if k.startswith("optimizer_ids_kernel")}
    assert len(opt) >= 1
    for name, (_, longest) in opt.items():
        # what is left are the ranked / hot-range paths' pointer-chasing loops (<= 3); the id scan itself had 8
        assert longest <= 3, (name, longest)
    r = audit.chains(_isa(tmp_path, CSRC / "route.hip"))
    routes = {k: v for k, v in r.items() if k.startswith("route_kernel")}
    assert len(routes) >= 1
    assert all(longest < 2 for _, longest in routes.values()), routes


def test_fused_tower_input_tile_has_no_chain_of_single_dependent_loads(tmp_path):
    audit = _mod("audit_serial_loads")
    c = audit.chains(_isa(tmp_path, CSRC / "tower.hip"))
    tw = {k: v for k, v in c.items() if k.startswith("tower_fwd2_kernel")}
    assert len(tw) >= 1
    assert all(longest < 2 for _, longest in tw.values()), tw


def _barrier_report(path):
    audit = _mod("audit_barriers")
    lines = open(path).read().split("\n")
    rep = {}
    for name, body in audit.kernels(lines):
        r = audit.audit(body)
        r["flag"] = bool(r["vector"] or r["unknown"] or r["masked"] or (r["in_loop"] and not r["scalar"]))
        rep[name] = r
    return rep


@pytest.mark.parametrize("loop_form", ["plain", "lambda"])
def test_every_scorer_barrier_loop_closes_on_scalar_control(tmp_path, loop_form):
    """All score_kernel instantiations (exact f32 and bf16x3; FWD / FUSED / BWD / RANK / FUSED_S / BWD_S x ids x hard negatives),
    in the shipping loop form and in the lambda form r02's note blamed (-DTT_LOOP_LAMBDA=1)."""
    extra = ("-DTT_LOOP_LAMBDA=1",) if loop_form == "lambda" else ()
    rep = _barrier_report(_isa(tmp_path, CSRC / "score.hip", extra))
    assert len(rep) >= 100, len(rep)                       # (112-128 instantiations, by build flags)
    in_loop = [n for n, r in rep.items() if r["in_loop"]]
    assert len(in_loop) == len(rep), "every scorer kernel has its per-tile barrier inside the tile loop"
    bad = {n: r for n, r in rep.items() if r["flag"]}
    assert not bad, {re.sub(r"^_ZN\d+_GLOBAL__N_1", "", n)[:60]: r for n, r in bad.items()}


def test_barrier_audit_flags_a_loop_on_a_per_lane_trip_count(tmp_path):
    """The negative control: without it a green audit could mean 'the script finds nothing, ever'."""
    rep = _barrier_report(_isa(tmp_path, AUDIT / "audit_negative_control.hip"))
    assert len(rep) == 1
    (r,) = rep.values()
    assert r["in_loop"] >= 1 and r["flag"] and (r["vector"] or r["masked"]), r


def _resources(tmp_path, name):
    """{demangled kernel: (vgprs, agprs, scratch bytes per lane, occupancy)} from hipcc's kernel-resource-usage remarks."""
    if not pathlib.Path(HIPCC).exists():
        pytest.skip("hipcc not available")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT / 'include'}",
                        "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "-o", str(tmp_path / f"{name}.o"),
                        str(CSRC / f"{name}.hip")], check=True, capture_output=True, text=True, timeout=900)
    rows, cur = [], {}
    for line in r.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            if cur:
                rows.append(cur)
            cur = {"name": m.group(1)}
        for key, tag in (("VGPRs", "v"), ("AGPRs", "a"), (r"ScratchSize \[bytes/lane\]", "scr"), (r"Occupancy \[waves/SIMD\]", "occ")):
            m = re.search(r" " + key + r": (\d+)", line)
            if m and cur:
                cur[tag] = int(m.group(1))
    if cur:
        rows.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for x, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        out[re.sub(r"\(.*", "", n).replace("void ", "")] = (x.get("v"), x.get("a"), x.get("scr"), x.get("occ"))
    return out


def test_no_kernel_of_the_train_step_uses_scratch(tmp_path):
    """r03 left 40-132 B of scratch per lane in the exact-f32 dim-256 online-softmax kernels with accidental-hit ids / hard
    negatives (cfg5 with remove_accidental_hits runs one of them) and 20 B in a dim-128 gradient kernel; r04 keeps the last
    k-groups of their stationary fragment in LDS.  Every kernel of every translation unit the trainers launch must be free of
    scratch - except ONE instantiation only the custom op can reach (bf16x3, dim 128, ids AND hard negatives: 44 B), which is
    pinned here so that it does not grow unnoticed."""
    known = {"score_kernel<128, 4, true, true, 8, 1>": 44}
    seen = 0
    for tu in ("score", "sparse", "gemm", "tower", "sort"):
        for name, (v, a, scr, occ) in _resources(tmp_path, tu).items():
            seen += 1
            assert scr is not None, name
            if name in known:
                assert scr <= known[name], (name, scr)
            elif tu == "sort" and "rocprim" in name:
                continue                                     # the vendor fallback beyond 262,144 ids (outside every BASELINE config)
            else:
                assert scr == 0, (tu, name, v, a, scr, occ)
    assert seen >= 150
