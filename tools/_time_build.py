import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import fmx_import
fmx = fmx_import.load().fmx
import workloads
c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(5_000_000, 1000, 25)
t = time.time(); ix = fmx.Index.create(); ix.add_ids(c_off, c_tok, vs); print("add_ids %.2fs" % (time.time() - t))
t = time.time(); ix.sort(0); print("sort %.2fs" % (time.time() - t))
t = time.time(); ix.finalize(); print("finalize (derive) %.2fs" % (time.time() - t))
t = time.time(); ix.to_device(0); print("to_device %.2fs" % (time.time() - t))
t = time.time(); ix.save("/tmp/c3.fmi"); print("save %.2fs" % (time.time() - t), os.path.getsize("/tmp/c3.fmi") / 1e9, "GB")
t = time.time(); ix2 = fmx.Index.load("/tmp/c3.fmi"); print("load %.2fs" % (time.time() - t))
t = time.time(); ix2.finalize(); print("finalize %.2fs" % (time.time() - t))
t = time.time(); ix2.to_device(0); print("to_device %.2fs" % (time.time() - t))
