"""Host-side profile of the validation loop (main_dist --only_val on synthetic batches): where the milliseconds per batch go."""
import cProfile, importlib, io, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
m = importlib.import_module("vognet-pytorch_amd.main_dist")
d = tempfile.mkdtemp()
kw = {"mdl.name": "vog", "ds.conc_type": "spat", "mdl.obj_tx.use_rel": True, "mdl.mul_tx.use_rel": True, "train.bsv": 4, "misc.tmp_path": d,
      "hip.batch_requests": int(os.environ.get("BR", "1"))}
m.main_dist("w", only_val=True, synthetic_batches=8, **kw)
if os.environ.get("NOPROF") == "1":
    m.main_dist("u", only_val=True, synthetic_batches=128, **kw); m.main_dist("u", only_val=True, synthetic_batches=128, **kw); sys.exit(0)
pr = cProfile.Profile(); pr.enable()
m.main_dist("v", only_val=True, synthetic_batches=128, **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callees("eval_vsrl_corr.py:.*forward"); print(s.getvalue()[:9000])
