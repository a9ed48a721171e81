import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, pkgload
from oracle import oracle as orc
from cases import CASES
pkg = pkgload.load(); ghf = pkg.ghf
ctx = ghf.Context(0)
data = CASES["uniform_1m"]()
d_in = torch.from_numpy(data).cuda()
d_out, nbytes, d_code = ctx.compress(d_in)
ctx.sync()
nb = int(nbytes.item()); got = d_out[:nb].cpu().numpy(); ref = orc.compress(data)
print(nb, ref.size)
d = np.nonzero(got[:min(nb, ref.size)] != ref[:min(nb, ref.size)])[0]
print("ndiff", d.size, d[:20], d[-5:] if d.size else None)
hs = 1040 + 8 * 9
print("chunk of first diffs (4 KiB chunks ~ 4100 bytes out):", ((d[:20] - hs) / 4100.5).astype(int))
