import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "scratch")
from r5_quant_envelope import run, bf, h
E = None
S = [("all f16", {"tx": h}),
     ("enc exact", {"tx": h, "enc": E}),
     ("lstm exact", {"tx": h, "lstm": E}),
     ("head exact", {"tx": h, "head": E}),
     ("enc+lstm exact", {"tx": h, "enc": E, "lstm": E}),
     ("enc+lstm+qk+proj exact", {"tx": h, "enc": E, "lstm": E, "tx.qk": E, "tx.proj": E}),
     ("enc+lstm+qk+proj split", {"tx": h, "enc": "split", "lstm": "split", "tx.qk": "split", "tx.proj": "split"}),
     ("enc+qk+proj split", {"tx": h, "enc": "split", "tx.qk": "split", "tx.proj": "split"}),
     ("enc+qk+proj+head split", {"tx": h, "enc": "split", "tx.qk": "split", "tx.proj": "split", "head": "split"}),
     ("all but ffn/wo exact", {"tx": E, "tx.ffn": h, "tx.wo": h, "enc": E, "lstm": E, "head": E}),
     ]
import r5_quant_envelope
for n in sys.argv[1:] or ["full/cfg2_sharp16"]:
    run(n, S)
