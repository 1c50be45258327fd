"""CPU-side checks: the C-ABI library loads and exports every symbol the header
declares; host logic (config, selector, sharding, record layout, CLI parsing)."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = importlib.import_module("vognet-pytorch_amd.lib")
ec = importlib.import_module("vognet-pytorch_amd.extended_config")
sel = importlib.import_module("vognet-pytorch_amd.mdl_selector")
D = importlib.import_module("vognet-pytorch_amd.dist")
synth = importlib.import_module("vognet-pytorch_amd.synth")
main_dist = importlib.import_module("vognet-pytorch_amd.main_dist")


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "vog_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vog_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
    lib = L.load()
    names = _header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libvog_hip.so does not export {n}"
        assert n in L.SYMBOLS, f"ctypes binding missing for {n}"
    assert set(L.SYMBOLS) == set(names)
    assert lib.vog_version() == 1
    assert lib.vog_pred_record_bytes(4, 5, 10) == 6800     # 6800 B / query
    assert lib.vog_pred_record_bytes(1, 5, 10) == 2000


def test_ctx_weight_names_match_reference_state_dict_keys():
    """vog_ctx_create is host-only: the expected weight list must be exactly the
    keys the reference forward reads (SURVEY 8(b) Checkpoint)."""
    lib = L.load()
    eng = importlib.import_module("vognet-pytorch_amd.engine")
    cfg = ec.get_default_cfg()
    cfg.mdl.obj_tx.use_rel = True
    comm = {"vocab_size": 5000, "num_prop_per_frm": 5}
    d = eng.model_desc_from_cfg(cfg, comm)
    h = ctypes.c_void_p()
    assert lib.vog_ctx_create(ctypes.byref(d), ctypes.byref(h)) == 0
    n = lib.vog_ctx_num_weights(h)
    exp = {lib.vog_ctx_weight_name(h, i).decode(): lib.vog_ctx_weight_numel(h, i) for i in range(n)}
    sd = synth.init_state_dict(cfg, 5000)
    unused = {k for k in sd if k.startswith(("srl_simple_lin", "lin_tmp"))}
    assert set(exp) == set(sd) - unused
    for k, v in exp.items():
        assert v == sd[k].size, k
    # wrong size / unknown name are rejected with a message
    a = np.zeros(3, np.float32)
    assert lib.vog_ctx_set_weight(h, b"lin2.2.bias", a.ctypes.data, 3) != 0
    assert b"numel" in lib.vog_last_error()
    assert lib.vog_ctx_set_weight(h, b"nope.weight", a.ctypes.data, 3) != 0
    # DDP prefix and legacy LayerNorm names are accepted
    g = np.ones(512, np.float32)
    assert lib.vog_ctx_set_weight(h, b"module.obj_txf.encoder.layers.0.selfattn.layernorm.gamma",
                                  g.ctypes.data, 512) == 0
    assert lib.vog_ctx_finalize(h) != 0 and b"missing weight" in lib.vog_last_error()
    lib.vog_ctx_destroy(h)


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    eng = importlib.import_module("vognet-pytorch_amd.engine")
    cfg = ec.get_default_cfg()
    with pytest.raises(L.VogError):
        eng.VogEngine(cfg, {"vocab_size": 10, "num_prop_per_frm": 5})


def test_selector_contract():
    cfg = ec.get_default_cfg()
    for ct in ("sep", "svsq", "temp", "spat"):
        for m in ("igrnd", "vgrnd", "vog"):
            cfg.ds.conc_type, cfg.mdl.name = ct, m
            r = sel.get_mdl_loss_eval(cfg)
            assert set(r) == {"mdl", "loss", "eval"}
            want = {"igrnd": "ImgGrnd", "vgrnd": "VidGrnd", "vog": "VOG"}[m] + "_" + \
                   ("SEP" if ct in ("sep", "svsq") else ct.upper())
            assert r["mdl"].__name__ == want
    cfg.mdl.name = "nope"
    with pytest.raises(NotImplementedError):
        sel.get_mdl_loss_eval(cfg)
    cfg.mdl.name, cfg.ds.conc_type = "vog", "nope"
    with pytest.raises(NotImplementedError):
        sel.get_mdl_loss_eval(cfg)


def test_model_state_dict_keys_and_load():
    cfg = ec.get_default_cfg()
    cfg.mdl.rnn.rnn_size = 32
    cfg.mdl.input_encoding_size = 16
    comm = {"vocab_size": 50, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": 5}
    mdl = sel.get_mdl_loss_eval(cfg)["mdl"](cfg=cfg, comm=comm)
    keys = set(mdl.state_dict().keys())
    assert keys == set(synth.init_state_dict(cfg, 50).keys())
    assert "obj_txf.encoder.layers.0.selfattn.layer.wq.weight" in keys
    assert "lstm_encoder.lstm.weight_hh_l1_reverse" in keys
    sd = {("module." + k): v for k, v in mdl.state_dict().items()}
    g = sd.pop("module.mult_txf.encoder.layers.0.feedforward.layernorm.weight")
    sd["module.mult_txf.encoder.layers.0.feedforward.layernorm.gamma"] = g
    mdl.load_state_dict(sd, strict=True)


def test_cfg_update_from_dict_checks():
    cfg = ec.get_default_cfg()
    ec.update_from_dict(cfg, {"mdl.obj_tx.use_rel": "True", "train.lr": "5e-4", "ds.conc_type": "temp"})
    assert cfg.mdl.obj_tx.use_rel is True and cfg.train.lr == 5e-4 and cfg.ds.conc_type == "temp"
    with pytest.raises(AssertionError):
        ec.update_from_dict(cfg, {"mdl.no_such_key": 1})
    with pytest.raises(AssertionError):
        ec.update_from_dict(cfg, {"mdl.obj_tx.n_layers": "three"})
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.mdl.name = "igrnd"


def test_cli_parse():
    uid, kw = main_dist.parse_argv(["exp1", "--ds.conc_type=spat", "--mdl.obj_tx.use_rel=True",
                                    "--local_rank", "3", "--only_val"])
    assert uid == "exp1" and kw == {"ds.conc_type": "spat", "mdl.obj_tx.use_rel": "True",
                                    "local_rank": "3", "only_val": "True"}


def test_shard_matches_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    for n, w in ((10, 4), (16, 8), (7, 2), (5, 8)):
        for r in range(w):
            ds = DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False)
            total = ds.total_size
            idx = list(range(n))
            idx += idx[: total - n] if total - n <= n else (idx * (total // n + 1))[: total - n]
            ref = idx[ds.num_samples * r: ds.num_samples * (r + 1)]
            assert D.shard_indices(n, r, w) == ref, (n, w, r)


def test_ctypes_structs_match_the_c_header(tmp_path):
    """sizeof and the offset of the LAST member of every struct of include/vog_hip.h, as gcc lays
    them out, against the ctypes mirrors in lib.py (a missing trailing field would make the library
    read garbage past the Python struct)."""
    import ctypes as C
    import shutil
    import subprocess
    L = importlib.import_module("vognet-pytorch_amd.lib")
    pairs = {"vog_gemm_args": L.GemmArgs, "vog_splitk_prob": L.SplitkProb, "vog_qkv_args": L.QkvArgs,
             "vog_qkvcomb_args": L.QkvCombArgs, "vog_attn_args": L.AttnArgs,
             "vog_attn_struct_args": L.AttnStructArgs, "vog_visprep_args": L.VisprepArgs,
             "vog_lstm_step_args": L.LstmStepArgs, "vog_lstm_layer_args": L.LstmLayerArgs,
             "vog_vislang_args": L.VislangArgs, "vog_score_args": L.ScoreArgs,
             "vog_predcmp_args": L.PredcmpArgs, "vog_pred_args": L.PredArgs, "vog_tx_tail_args": L.TxTailArgs, "vog_encoder_layer_args": L.EncoderLayerArgs, "vog_visenc_args": L.VisencArgs, "vog_loss_args": L.LossArgs, "vog_tail_bwd_args": L.TailBwdArgs, "vog_attn_f32_args": L.AttnF32Args, "vog_linear_f32_args": L.LinearF32Args, "vog_lang_f32_args": L.LangF32Args, "vog_assemble_args": L.AssembleArgs, "vog_copy_seg": L.CopySeg,
             "vog_model_desc": L.ModelDesc, "vog_batch": L.Batch}
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "vog_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        last = cls._fields_[-1][0]
        src.append(f'  printf("{cname} %zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));')
    src += ['  return 0;', '}']
    cfile = tmp_path / "abi.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "abi"
    subprocess.run([gcc, "-I", os.path.join(root, "include"), str(cfile), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        cname, size, off = line.split()
        cls = pairs[cname]
        assert C.sizeof(cls) == int(size), (cname, C.sizeof(cls), size)
        assert getattr(cls, cls._fields_[-1][0]).offset == int(off), (cname, off)


def test_training_switch_is_per_thread_not_process_state():
    """vog_train_set_int("bf16_gemm") belongs to the calling thread (host-only check; the device side of it is
    tests/test_gpu_surface.py::test_two_trainers_with_different_gemm_settings_in_one_process)."""
    import threading
    lib = L.load()

    def get():
        v = ctypes.c_int32(-1)
        assert lib.vog_train_get_int(b"bf16_gemm", ctypes.byref(v)) == 0
        return v.value

    assert lib.vog_train_set_int(b"bf16_gemm", 1) == 0 and get() == 1
    seen = {}

    def other():
        seen["before"] = get()
        lib.vog_train_set_int(b"bf16_gemm", 0)
        seen["after"] = get()

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == {"before": 0, "after": 0}
    assert get() == 1                      # the other thread's write did not reach this one
    assert lib.vog_train_set_int(b"bf16_gemm", 0) == 0 and get() == 0
    assert lib.vog_train_set_int(b"no_such_switch", 1) != 0


def test_synthetic_loader_keeps_the_short_batch_last_on_its_rank():
    """shard_indices wraps around, so on world = 4 with 10 validation batches rank 3 owns batches [9, 0, 1] - the
    short tail batch must still be the LAST one it sees (a loader yields its tail last; the evaluator's ring is sized by
    cfg.train.bsv, never by a short first batch)."""
    cfg = ec.get_default_cfg()
    comm = {"vocab_size": 200, "detect_size": 431, "itod": {}, "wtoi": {"UNK": 1}, "num_prop_per_frm": ec.num_prop_per_frm(cfg)}
    bs = int(cfg.train.bsv)
    seen_short = 0
    for rank in range(4):
        idx = list(D.shard_indices(10, rank, 4))
        if 9 in idx and idx[0] == 9:
            dl = main_dist.synthetic_loader(cfg, comm, 10, rank, 4)
            sizes = [int(b["num_cmp_msk"].shape[0]) for b in dl]
            assert sizes[0] == bs and sizes[-1] == bs - 1 and sizes.count(bs - 1) == 1, sizes
            seen_short += 1
    assert seen_short >= 1


def test_cpulist_parser_and_numa_binding_without_a_gpu():
    """dist.bind_host_to_device_node: the sysfs cpulist syntax, and no GPU / no topology -> None with the affinity untouched."""
    D = importlib.import_module("vognet-pytorch_amd.dist")
    assert D._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert D._parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    assert D.bind_host_to_device_node(0) is None
    assert os.sched_getaffinity(0) == before


def test_device_prefetcher_is_transparent_on_cpu():
    """Without a GPU the prefetcher hands the loader's batches through, in order."""
    import torch
    dls = importlib.import_module("vognet-pytorch_amd.dat_loader_simple")
    dl = [{"a": torch.full((2, 3), i)} for i in range(5)]
    got = [(d["a"][0, 0].item(), h is dl[i]) for i, (d, h) in enumerate(dls.DevicePrefetcher(dl, "cpu", depth=2, hold=3))]
    assert got == [(i, True) for i in range(5)]
