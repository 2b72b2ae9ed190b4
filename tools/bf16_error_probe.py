import sys, glob, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_parity import _model
from conftest import load_golden
dev = torch.device('cuda', 0)
for name in ["cell_dari_tult_B256_T3_F80.npz", "cell_dari_tult_B256_T3_F64.npz", "cell_dari_tult2_B3_T7_F80.npz", "cell_dari_tult_B4_T3_F64.npz"]:
    g = load_golden(name)
    F = g["x"].shape[2]
    m = _model(dev, F // 16, "dari_tult2" if "dari_tult2" in name else "dari_tult")
    m.conv_precision = "bf16"
    out, hx = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["hx0"]).to(dev))
    err = out.cpu().numpy() - g["out"]
    print(name, "max-abs", np.abs(err).max(), "rel-rms", np.sqrt(np.mean(err**2))/np.sqrt(np.mean(g["out"]**2)), "max|out|", np.abs(g["out"]).max(), "hx err", np.abs(hx.cpu().numpy()-g["hx1"]).max())
# realistic input: hx from zeros, chained hops
g = load_golden("cell_dari_tult_chain20_F80.npz")
m = _model(dev, 5); m.conv_precision = "bf16"
hx = None; worst = 0; num = 0; den = 0
for h in range(20):
    o, hx = m(torch.from_numpy(g["x"][h]).to(dev), hx)
    e = o.cpu().numpy() - g["out"][h]; worst = max(worst, np.abs(e).max()); num += (e**2).sum(); den += (g["out"][h]**2).sum()
print("chain20 max-abs", worst, "rel-rms", np.sqrt(num/den))
