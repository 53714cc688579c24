"""CPU tests of the host-side mirrors of the reference interface (no GPU, no kernels): structure / state_dict layout,
analytic MACs and penalties, schedules, PruneInfo protocol, config loader, C-ABI symbol table, reference known answers."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _tables():
    return torch.load(os.path.join(GOLD, "tables.pt"), weights_only=False)


@pytest.mark.parametrize("name,key", [("atomnas_c", "atomnas_c_supernet"), ("atomnas_a", "atomnas_a_supernet")])
def test_supernet_structure_matches_reference(name, key):
    """state_dict keys (order!) and shapes, parameter / tensor counts, MAC stamps and L1 penalties of the full supernets."""
    from atomnas_amd import configs
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils import model_profiling as mp
    from atomnas_amd.utils import prune
    t = _tables()[name]
    model = ms.Model(**configs.model_kwparams(key), input_size=224)
    sd = model.state_dict()
    assert list(sd.keys()) == t["keys"]
    assert [tuple(v.shape) for v in sd.values()] == t["shapes"]
    assert len(list(model.parameters())) == t["n_tensors"]
    macs, params = mp.model_profiling(model, 224, 224, verbose=False)
    assert macs == t["n_macs"] and params == t["n_params"]
    assert [b.n_macs for b in model.get_named_block_list().values()] == t["block_macs"]
    pinfo = prune.get_bn_to_prune(model, {"bn_prune_filter": "expansion_only_skip_expand1"}, verbose=False)
    assert pinfo.weight == t["names"]
    assert pinfo.penalty == t["penalties"]          # python doubles: bit-exact
    assert pinfo.get_info_list("per_channel_flops") == t["pcf"]


def test_cfg1_mobilenet_v2_plumbing():
    """BASELINE config 1 (MobileNetV2 1.0, CPU plumbing): yaml -> kwargs -> model -> MAC stamps -> prune info."""
    os.environ.setdefault("ARNOLD_OUTPUT", "/tmp/atomnas_out")
    os.environ.setdefault("DATA_LMDB", "/tmp/none")
    from atomnas_amd.utils import config, model_profiling as mp
    from atomnas_amd.models import mobilenet_supernet as ms
    flags = config.load_app(["app:" + os.path.join(ROOT, "apps/mobilenet/mobilenet_v2_mnas.yml"), "--per_gpu_batch_size", "2"])
    assert flags.per_gpu_batch_size == 2 and flags.model_kwparams.batch_norm_momentum == 0.01 and flags.optimizer == "rmsprop"
    assert flags.prune_params.method is None and flags.model_shrink_delta_flops == 1e100   # defaults from slimming_base.yml
    model = ms.Model(**flags.model_kwparams, input_size=flags.image_size)
    macs, params = mp.model_profiling(model, 224, 224, verbose=False)
    assert sum(p.numel() for p in model.parameters()) == 3504872 and 299e6 < macs < 302e6   # MobileNetV2-1.0: 3.50 M parameters, ~300 M MACs
    x = torch.zeros(2, 3, 224, 224)
    with pytest.raises(Exception):   # the product has no CPU path: it must fail loudly, not fall back
        model(x)


def test_config_loader_semantics(tmp_path):
    from atomnas_amd.utils import config
    os.environ["ATOMNAS_TEST_DIR"] = str(tmp_path)
    (tmp_path / "base.yml").write_text("a: 1\nb: {c: 2, d: [1, 2]}\nflag: False\npath: ${ATOMNAS_TEST_DIR}/x\n")
    (tmp_path / "mid.yml").write_text("_default: !include ./base.yml\na: 5\n'b.c': 7\n")
    (tmp_path / "top.yml").write_text("_default: !include ${ATOMNAS_TEST_DIR}/mid.yml\nextra: !include ./base.yml\n")
    f = config.load_app(["app:" + str(tmp_path / "top.yml"), "--a", "9", "--b.c", "11", "--flag", "False"])
    assert f.a == 9 and f.b.c == 11 and f.b.d == [1, 2] and f.extra.a == 1
    assert f.flag is True            # type(old)(val): bool('False') is True, as in the reference
    assert f.path == str(tmp_path) + "/x"
    with pytest.raises(RuntimeError):
        config.load_app(["app:" + str(tmp_path / "top.yml"), "--missing", "1"])
    with pytest.raises(RuntimeError):
        config.load_app([str(tmp_path / "top.yml")])


def test_search_configs_resolve_like_the_reference():
    os.environ.setdefault("ARNOLD_OUTPUT", "/tmp/atomnas_out")
    os.environ.setdefault("DATA_LMDB", "/tmp/none")
    from atomnas_amd import configs
    from atomnas_amd.utils import config
    f = config.load_app(["app:" + os.path.join(ROOT, "apps/slimming/shrink/atomnas_c.yml")])
    hp = configs.SEARCH_HPARAMS
    for k in ("optimizer", "momentum", "alpha", "epsilon", "eps_inside_sqrt", "weight_decay", "weight_decay_method", "base_lr",
              "base_total_batch", "lr_scheduler", "exp_decaying_lr_gamma", "exp_decay_epoch_interval", "label_smoothing",
              "moving_average_decay", "moving_average_decay_base_batch", "num_epochs", "random_seed", "model_shrink_threshold",
              "model_shrink_delta_flops"):
        assert f[k] == hp[k], k
    assert dict(f.prune_params) == hp["prune_params"]
    kw = dict(f.model_kwparams)
    ref = configs.model_kwparams("atomnas_c_supernet")
    assert {k: kw[k] for k in ref} == ref
    fa = config.load_app(["app:" + os.path.join(ROOT, "apps/slimming/shrink/atomnas_a.yml")])
    assert fa.prune_params.rho == 1.8e-4 and fa.model_kwparams.input_channel == 16 and fa.bn_calibration_steps == 10


def test_schedules_known_answers():
    """rho / lr / EMA-decay known answers of the reference ([probed] values in SURVEY.md section 8c; tests/utils/prune_test.py:38-49)."""
    import types
    from atomnas_amd.utils import optim, prune
    rs = prune.get_rho_scheduler(dict(rho=1.0, epoch_free=1, epoch_warmup=3, scheduler="linear", stepwise=True), 2)
    assert [rs(i) for i in range(15)] == [0, 0, 0, 0.25, 0.5, 0.75] + [1.0] * 9
    t = _tables()
    rs = prune.get_rho_scheduler(dict(rho=1e-4, epoch_free=0, epoch_warmup=25, scheduler="linear", stepwise=True), 626)
    assert [rs(i) for i in t["rho"]["idx"]] == t["rho"]["val"]
    flags = types.SimpleNamespace(lr=0.128, base_lr=0.016, _steps_per_epoch=626, lr_scheduler="exp_decaying", exp_decay_epoch_interval=2.4,
                                  exp_decaying_lr_gamma=0.97, num_epochs=350)
    flags.get = lambda k, d=None: {"lr_stepwise": False, "epoch_warmup": 5}.get(k, d)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.128)
    lam = optim.get_lr_scheduler(opt, flags).lr_lambdas[0]
    for i, v in zip(t["lr"]["idx"], t["lr"]["val"]):
        assert abs(0.128 * lam(i) - v) < 1e-15
    assert abs(0.128 * lam(3131) - 0.1204352) < 1e-9 and abs(0.128 * lam(626) - 0.0384) < 1e-12
    assert optim.ExponentialMovingAverage.adjust_momentum(0.9999, 2.0) == t["ema_decay"]["adjusted"]
    ema = optim.ExponentialMovingAverage(0.99994999875)
    assert [ema.momentum_at(n) for n in (1, 10, 100, 100000, 1000000)] == t["ema_decay"]["sched"]


def test_prune_info_protocol():
    """PruneInfo rename / drop bookkeeping (utils/prune.py:40-86; tests/models/mobilenet_base_test.py:61-108)."""
    from atomnas_amd.utils.prune import PruneInfo
    pi = PruneInfo(["a.weight", "b.weight", "c.weight"], [1.0, 2.0, 3.0])
    pi.add_info_list("per_channel_flops", [10, 20, 30])
    pi.compress_start()
    assert pi.compress_check_exist({"var_old_name": "b.weight"}) and not pi.compress_check_exist({"var_old_name": "zz"})
    pi.compress_mask({"var_old_name": "c.weight", "var_new_name": "b.weight2"})
    assert pi.compress_drop({"var_old_name": "a.weight"})["penalty"] == 1.0
    assert pi.weight == ["b.weight", "b.weight2"] and pi.penalty == [2.0, 3.0]
    with pytest.raises(RuntimeError):
        pi.compress_mask({"var_old_name": "b.weight2", "var_new_name": "x"})   # already moved in this round
    pi.compress_start()
    pi.compress_mask({"var_old_name": "b.weight2", "var_new_name": "b.weight"})  # rename onto an existing, not yet moved key
    assert pi.weight == ["b.weight"] and pi.penalty == [3.0]


def test_masks_and_l1_subgradient_known_answers():
    """tests/utils/prune_test.py:54-64 (threshold masks) and :23-36 (L1 sub-gradient rho*penalty*sign(gamma), sign(0) = 0) on the
    oracle (the HIP versions are checked against the oracle on the GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import atomnas_oracle as orc
    from atomnas_amd.utils import prune
    x = [torch.tensor(v, dtype=torch.float32) for v in ([1, 2, 5], [3, 6, 0, 1.1])]
    exp = [torch.tensor([False, True, True]), torch.tensor([True, True, False, False])]
    for got in (prune.cal_mask_network_slimming_by_threshold(x, 1.5), [orc.alive_mask(t, 1.5) for t in x]):
        assert all(torch.equal(a, b) for a, b in zip(got, exp))
    pinfo = prune.PruneInfo(["one", "two"], [0, 1])
    pinfo.add_info_list("per_channel_flops", [3, 5])
    mask, thr = prune.cal_mask_network_slimming_by_flops(x, pinfo, 12)
    assert abs(thr - 1.1) < 1e-6 and all(torch.equal(a, b) for a, b in zip(mask, exp))
    mask, thr = prune.cal_mask_network_slimming_by_flops(x, pinfo, 13)
    assert thr == 2 and torch.equal(mask[0], torch.tensor([False, False, True]))
    pinfo.add_info_list("mask", mask)
    assert prune.cal_pruned_flops(pinfo)[0] >= 13
    g = torch.tensor([0.0, -2.0, 3.0], requires_grad=True)
    orc.bn_l1_loss([g], [2.0], 0.5).backward()
    assert g.grad.tolist() == [0.0, -1.0, 1.0]


def test_ce_label_smooth_and_ema_known_answers():
    """tests/utils/optim_test.py:14-32 (CE-smooth = 400*L/3 ... restated: uniform logits) and EMA update values."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import atomnas_oracle as orc
    logits = torch.zeros(2, 4)
    loss = orc.ce_label_smooth(logits, torch.tensor([1, 3]), 0.1)
    assert torch.allclose(loss, torch.full((2,), float(torch.log(torch.tensor(4.0)))))   # uniform prediction: log K whatever eps
    logits = torch.tensor([[10.0, 0.0, 0.0]])
    l0 = orc.ce_label_smooth(logits, torch.tensor([0]), 0.0)
    l1 = orc.ce_label_smooth(logits, torch.tensor([0]), 0.3)
    lp = torch.log_softmax(logits, 1)[0]
    assert abs(float(l1) - float(-(0.7 + 0.1) * lp[0] - 0.1 * lp[1] - 0.1 * lp[2])) < 1e-6 and float(l0) < float(l1)
    s = torch.tensor([1.0, 2.0])
    orc.ema_update(s, torch.tensor([3.0, 4.0]), orc.ema_decay(0.9, 0))     # min(0.9, 1/10) = 0.1
    assert torch.allclose(s, torch.tensor([0.1 * 1 + 0.9 * 3, 0.1 * 2 + 0.9 * 4]))
    assert orc.ema_decay(0.9, None) == 0.9 and orc.ema_decay(0.9, 1000) == 0.9


def test_c_abi_library_exports_every_declared_symbol():
    """libatomnas_hip.so loads without a GPU and exports exactly the entry points include/atomnas_hip.h declares."""
    from atomnas_amd import _lib
    from atomnas_amd import build
    build.build_library()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "atomnas_hip.h")).read()
    declared = set(re.findall(r"\b(atomnas_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    for sym in declared:
        assert hasattr(lib, sym), "missing symbol " + sym
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert lib.atomnas_abi_version() == _lib.ABI_VERSION == 9
    assert isinstance(lib.atomnas_last_error(), bytes)


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "atomnas_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "atomnas_oracle" not in src and "import oracle" not in src, os.path.join(base, f)
    for f in ("train.py", "common.py"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_e2e_test_config_resolves_on_cpu(tmp_path):
    """The config of the GPU entry-point test (tests/test_train_entry_gpu.py) must load without a GPU: include chain, dotted
    overrides and environment expansion."""
    os.environ["ATOMNAS_E2E_DIR"] = str(tmp_path)
    os.environ.setdefault("ARNOLD_OUTPUT", str(tmp_path))
    os.environ.setdefault("DATA_LMDB", "/tmp/none")
    from atomnas_amd.utils import config
    flags = config.load_app(["app:" + os.path.join(ROOT, "tests", "data", "tiny_search.yml")])
    assert flags.num_epochs == 2 and flags.max_steps_per_epoch == 3 and flags.per_gpu_batch_size == 8
    assert flags.log_dir == str(tmp_path) and flags.use_distributed is False
    assert flags.prune_params.method == "network_slimming" and flags.prune_params.rho == 1e-4
    assert flags.model_kwparams.active_fn == "nn.ReLU" and flags.model_kwparams.batch_norm_momentum == 0.01
    assert flags.resume == "" and flags.optimizer == "rmsprop"


def test_init_weights_mnas_reproduces_the_reference_under_a_seed():
    """SURVEY section 8 row a7 (models/mobilenet_base.py:440-459): conv ~ N(0, sqrt(2/fan_out)) with fan_out = k*k for depthwise,
    BN (1, 0), classifier ~ U(+-1/sqrt(out_features)), zero bias.  Module construction order and the init calls consume torch's
    RNG exactly as the reference does, so under the same seed every tensor is IDENTICAL to the reference's (digests generated by
    tools/make_golden.py checkpoint from the reference itself)."""
    import math
    from kutil import check_digest
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    g = torch.load(os.path.join(ROOT, "tests", "golden", "checkpoint_ref.pt"), weights_only=False)
    torch.manual_seed(g["seed"])
    model = ms.Model(**g["kw"])
    model.apply(mb.init_weights_mnas)
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["init_digest"].keys())
    for k, d in g["init_digest"].items():
        check_digest(k, sd[k], d, rtol=1e-7)
    # and the rule itself on a large layer (statistical)
    conv = torch.nn.Conv2d(64, 64, 5, groups=64, bias=False)
    dense = torch.nn.Conv2d(32, 256, 1, bias=False)
    fc = torch.nn.Linear(512, 1000)
    for m in (conv, dense, fc):
        mb.init_weights_mnas(m)
    assert abs(float(conv.weight.std()) - math.sqrt(2.0 / 25)) < 0.02
    assert abs(float(dense.weight.std()) - math.sqrt(2.0 / 256)) < 0.005
    assert float(fc.weight.abs().max()) <= 1 / math.sqrt(1000) + 1e-7 and float(fc.bias.abs().max()) == 0.0


def test_ctypes_signatures_match_the_header_prototypes():
    """Every binding in atomnas_amd/_lib.py has the argument list of its prototype in include/atomnas_hip.h: same count, and per
    position pointer / 32-bit int / 64-bit long / float / double.  (ctypes would silently pass a mis-sized argument.)"""
    import ctypes
    from atomnas_amd import _lib
    header = open(os.path.join(ROOT, "include", "atomnas_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(?:int|const char\s*\*)\s+(atomnas_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", header, flags=re.S))
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_long: "long", ctypes.c_float: "float", ctypes.c_double: "double",
             ctypes.c_ulonglong: "long", ctypes.c_char_p: "ptr"}

    def kind_of(decl):
        d = decl.strip()
        if "*" in d:
            return "ptr"
        base = d.rsplit(None, 1)[0] if " " in d else d
        base = base.replace("const", "").strip()
        return {"int": "int", "long": "long", "unsigned long long": "long", "long long": "long", "float": "float", "double": "double"}[base]
    table = dict(_lib.SIGNATURES)
    table.update({k: v[1] for k, v in _lib.NO_STATUS.items()})
    assert set(table) == set(protos), set(table) ^ set(protos)
    for name, args in table.items():
        decl = protos[name].strip()
        want = [] if decl in ("", "void") else [kind_of(a) for a in decl.split(",")]
        got = [kinds[a] for a in args]
        assert got == want, (name, got, want)


def test_tools_and_entry_points_compile():
    """the experiment harnesses under tools/ and the repo-root entry points are importable Python (they run on the GPU box only)"""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, f) for f in ("bench.py", "train.py", "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, doraise=True)


def test_asm_loads_and_ring_waits_hold_on_the_isa():
    """ISA checks of the library's hand-synchronised memory operations (tools/check_asm_waits.py), on the device assembly the build
    keeps next to the objects (atomnas_amd/csrc/build/*.s, cached by source digest; generated here when missing -- minutes of hipcc
    the first time, a parse afterwards):
      * no instruction touches a register between the inline-asm load that writes it and the manual s_waitcnt lgkmcnt(0);
      * every counted `s_waitcnt vmcnt(N)` of the LDS-DMA rings (k_gemm_nt_sw, k_expand_bwd_s, k_gemm_tn3, k_gram_part) has at least N
        copies issued behind the end marker of the stage it waits for, on every path; no compiler-emitted vector-memory load sits in
        a ring loop."""
    import concurrent.futures
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc: the device assembly cannot be produced")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_waits as caw
    from atomnas_amd import build
    files = [os.path.join(ROOT, "atomnas_amd", "csrc", f) for f in build.ISA_CHECKED]
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(files)) as ex:
        asm = list(ex.map(build.assemble, files))
    findings, nloads, rings = [], 0, []
    for a in asm:
        f, n = caw.check(a)
        findings += f
        nloads += n
        f, r = caw.check_rings(a)
        findings += f
        rings += r
    assert not findings, "\n".join(findings[:20])
    assert nloads > 1000
    kinds = {r.split("ILi")[0].split("atomnas")[-1].lstrip("0123456789") for r in rings}
    assert {"k_gemm_nt_sw", "k_expand_bwd_s", "k_gemm_tn3", "k_gram_part"} <= kinds, kinds


def test_asm_wait_checker_follows_the_control_flow(tmp_path):
    """The checker itself (tools/check_asm_waits.py) on synthetic assembly: a consumer laid out BEFORE the block that waits is fine
    when every path to it passes the wait, a use on a path without the wait is a finding, a use inside inline asm is not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_waits as caw
    ok = """
_Z4goodv:
	s_cbranch_vccz .LBB0_3
.LBB0_1:
	v_add_f32_e32 v1, v10, v10
	s_endpgm
.LBB0_3:
	;;#ASMSTART
	ds_read_b64_tr_b16 v[10:11], v5
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	s_branch .LBB0_1
.Lfunc_end0:
"""
    bad = ok.replace("\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n", "")
    fo, fb = tmp_path / "ok.s", tmp_path / "bad.s"
    fo.write_text(ok)
    fb.write_text(bad)
    f, n = caw.check(str(fo))
    assert n == 1 and not f, f
    f, n = caw.check(str(fb))
    assert n == 1 and len(f) == 1 and "v_add_f32" in f[0], f


def test_asm_wait_checker_counted_lgkm_waits(tmp_path):
    """Counted `s_waitcnt lgkmcnt(N)` (k_gemm_nt_swg, round 6): LDS operations return in order, so the wait covers every asm load but the
    N operations issued last; a use of one of the last N is a finding, a compiler-issued LDS operation in between counts as one of
    them, and with a scalar-memory load in flight (out-of-order return) only lgkmcnt(0) covers anything."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_waits as caw

    def kern(body):
        return "_Z4kernv:\n" + body + "\ts_endpgm\n.Lfunc_end0:\n"
    rd = lambda r, a: "\t;;#ASMSTART\n\tds_read_b128 v[%d:%d], v%d\n\t;;#ASMEND\n" % (r, r + 3, a)
    wait = lambda n: "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(%d)\n\t;;#ASMEND\n" % n
    use = lambda r: "\tv_add_f32_e32 v1, v%d, v%d\n" % (r, r)
    cases = {
        "older_ok": (rd(10, 5) + rd(20, 6) + wait(1) + use(10), 0),
        "newest_in_flight": (rd(10, 5) + rd(20, 6) + wait(1) + use(20), 1),
        "compiler_lds_op_counts": (rd(10, 5) + "\tds_write_b32 v7, v8\n" + wait(1) + use(10), 0),
        "count_past_the_block": (rd(10, 5) + wait(2) + use(10), 1),
        "smem_in_flight": ("\t;;#ASMSTART\n\ts_load_dwordx2 s[4:5], s[0:1], 0x0\n\t;;#ASMEND\n" + rd(10, 5) + rd(20, 6) + wait(1) + use(10), 1),
        "zero_covers_all": (rd(10, 5) + rd(20, 6) + wait(0) + use(20) + use(10), 0),
    }
    for name, (body, nfind) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(kern(body))
        findings, _ = caw.check(str(f))
        assert len(findings) == nfind, (name, findings)


def test_ring_checker_counts_copies_behind_the_awaited_stage(tmp_path):
    """tools/check_asm_waits.check_rings on synthetic assembly: a ring of three stages of two copies with `s_waitcnt vmcnt(2)` in its loop
    (two stages ahead: exact) passes; the same loop with vmcnt(4) -- more than what was issued behind the awaited stage -- is a finding,
    and so is a compiler-emitted global load inside the ring loop."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_asm_waits as caw
    copy = "\t;;#ASMSTART\n\tglobal_load_lds_dwordx4 v1, off\n\t;;#ASMEND\n"
    mark = "\t;;#ASMSTART\n\t; atomnas_ring_stage_end\n\t;;#ASMEND\n"
    stage = copy + copy + mark
    def kernel(n, extra=""):
        return ("_Z4ringv:\n" + stage + stage + ".LBB0_1:                                ; =>This Inner Loop Header: Depth=1\n"
                "\t;;#ASMSTART\n\ts_waitcnt vmcnt(%d)\n\t;;#ASMEND\n\ts_barrier\n" % n + stage + extra +
                "\tv_add_f32_e32 v2, v3, v3\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n.Lfunc_end0:\n")
    for name, text, nfind, word in (("ok", kernel(2), 0, ""), ("early", kernel(4), 1, "before the stage has landed"),
                                    ("load", kernel(2, "\tglobal_load_dwordx4 v[4:7], v[8:9], off\n"), 1, "compiler-emitted")):
        f = tmp_path / (name + ".s")
        f.write_text(text)
        findings, rings = caw.check_rings(str(f))
        assert len(rings) == 1 and len(findings) == nfind, (name, findings, rings)
        if nfind:
            assert word in findings[0], findings
        else:
            assert "exact" in rings[0], rings
