"""Times zk_g2_add_batch's two kernels on n pairs of distinct points (run under rocprofv3 --kernel-trace --stats):
   python tools/time_g2_affine.py [log_n]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import zksnark_rs_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
ctx = zk.Context()
rng = np.random.default_rng(1)
# n distinct points: multiples of the generator by a table walk (k G, k = 1..) through add_batch itself
g = np.zeros((1, 16), dtype=np.uint64)
from tests.oracle_lib import load as load_oracle
orc = load_oracle()
base = orc.enc_base_g2().reshape(1, 16)
pts = base.copy()
while len(pts) < 2 * n:      # doubling trick: [P_i] -> [P_i] + [P_i + last]  (all multiples k G)
    shift = np.tile(pts[-1:], (len(pts), 1))
    pts = np.concatenate([pts, ctx.g2_add_batch(pts, shift)])
a, b = pts[:n].copy(), pts[n:2 * n].copy()
for opt in (0, 1):
    ctx.set_option("g2_affine", opt)
    ctx.g2_add_batch(a[:1024], b[:1024])
    t = time.time(); out = ctx.g2_add_batch(a, b); dt = time.time() - t
    print("g2_affine=%d  n=2^%d  wall %.3f s  sha %s" % (opt, log_n, dt, __import__("hashlib").sha256(out.tobytes()).hexdigest()[:16]))
