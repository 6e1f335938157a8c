"""Diagnostic: per-phase s_memtime stamps of workgroup 0 of the split-bf16 SRA forward (csrc/sra_attn.hip, sd_debug_sra_stamps).
    python tools/sra_stamps.py [D] [N] [heads]
Needs a DIAGNOSTIC build of the library (the product build has no stamp code and does not export sd_debug_sra_stamps):
    make -C segdistill_amd/csrc OUTDIR=../lib_stamps EXTRA=-DSD_SRA_STAMPS && SEGDISTILL_LIB=$PWD/segdistill_amd/lib_stamps/libsegdistill_hip.so python tools/sra_stamps.py
Prints, per wave, the cycle deltas between phase boundaries: staged | per tile: q planes, S^T issued, softmax done, PV issued, stored."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B, KV = 8, 256
dev = torch.device('cuda:0')
L = _lib.lib()
L.sd_debug_sra_stamps.restype, L.sd_debug_sra_stamps.argtypes = None, [C.c_void_p]
q = torch.randn(B, N, heads * D, device=dev)
kv = torch.randn(B, KV, 2 * heads * D, device=dev)
o, lse = torch.empty_like(q), torch.empty(B, heads, N, device=dev)
st = torch.zeros(8 * 32, dtype=torch.int64, device=dev)
run = lambda: _lib.check(L.sd_sra_fwd(q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), 0, B, N, KV, heads, D, D ** -0.5, None), 'fwd')
for _ in range(3):
    run()
torch.cuda.synchronize()
L.sd_debug_sra_stamps(st.data_ptr())
run()
torch.cuda.synchronize()
L.sd_debug_sra_stamps(None)
t = st.cpu().view(8, 32)
t0 = int(t[:, 0][t[:, 0] > 0].min())
for w in range(8):
    row = [int(x) for x in t[w] if x > 0]
    if not row:
        continue
    d = [row[0] - t0] + [row[i] - row[i - 1] for i in range(1, len(row))]
    print(f'wave {w}: start +{d[0]}  ' + ' '.join(str(x) for x in d[1:]) + f'   total {row[-1] - t0}')
