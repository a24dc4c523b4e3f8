import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from libultrahdr_amd.ultrahdr import UltraHdr
from libultrahdr_amd import capi as A, synth
ri = int(sys.argv[1])
u = UltraHdr()
device = "cuda:0"
w, h = 3840, 2160
sdr = synth.make_sdr_yuv420(w, h).to(device)
hq = [u.quant_table(95, False), u.quant_table(95, True), u.quant_table(95, True)]
hco = []
for c in range(3):
    rows, stride, wv = sdr.layout[c]
    hco.append(u.fdct_quant(sdr.plane_tensor(c), stride, wv // 8, rows // 8, hq[c]))
shp = [tuple(c.shape[:2]) for c in hco]
S = [(2, 2), (1, 1), (1, 1)]
stream = u.huffman_encode(hco, w, h, S, ri).clone()
for _ in range(12):
    u.huffman_decode(stream, shp, w, h, S, ri)
u.ctx.synchronize()
