"""How much of the score matrix the tile schedules process vs what the mask allows (CMU structure)."""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("mca-paper_amd")
cfg = P.config.cmu_model_config(batch_size=2, zorro=len(sys.argv) > 1 and sys.argv[1] == "mma")
model = P.MCA(**cfg)
st = model.structure
m = ~st.dense_attn_mask()         # dense_attn_mask: True = masked; m: True = allowed
N = m.shape[0]
print("N", N, "token dims", st.token_dims, "fusion", st.num_fusion_tokens, "allowed fraction", m.mean())
rows = np.add.reduceat(m.any(1).astype(int), [0])  # dummy
for bq, bk in [(128, 64), (64, 256), (64, 64), (64, 32), (32, 32), (32, 256), (128, 256), (64, 128)]:
    nq, nk = -(-N // bq), -(-N // bk)
    proc = 0
    for i in range(nq):
        for j in range(nk):
            if m[i * bq:(i + 1) * bq, j * bk:(j + 1) * bk].any(): proc += bq * bk
    print(f"tiles {bq:3d} q x {bk:3d} k: processed/N^2 = {proc / N / N:.3f}, processed/allowed = {proc / m.sum():.3f}")
# segment boundaries
b = np.concatenate([[0], np.cumsum(st.token_dims), [N]])
print("segment starts:", b.tolist())
# tiles that never cross a segment boundary (a tile = a segment and an offset inside it; the last tile of a segment is partial)
segs = [(int(b[i]), int(b[i + 1])) for i in range(len(b) - 1) if b[i + 1] > b[i]]
for bq, bk in [(128, 64), (64, 64), (64, 128), (32, 64)]:
    proc = 0
    for (q0, q1) in segs:
        for (k0, k1) in segs:
            if m[q0:q1, k0:k1].any():
                proc += (-(-(q1 - q0) // bq) * bq) * (-(-(k1 - k0) // bk) * bk)
    nqt = sum(-(-(q1 - q0) // bq) for q0, q1 in segs); nkt = sum(-(-(k1 - k0) // bk) for k0, k1 in segs)
    print(f"segment-aligned tiles {bq:3d} q x {bk:3d} k: {nqt} q tiles, {nkt} k tiles, processed/allowed = {proc / m.sum():.3f}")
