import sys, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("h2c", "/root/repo/tools/h2_check.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
core = m.core
g = torch.Generator().manual_seed(5)
Cin, Cout, grid = 256, 512, (50, 50, 4)
X, Y, Z = grid
x = torch.randn(1, Cin, X, Y, Z, generator=g)
w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.02
pc = core.PackedConv(w.to(m.dev), bn=m.bn_like(Cout, g).to(m.dev), ksize=3, stride=2, pad=1)
xr = m.rows_of(x)
core.CONV_ENGINE = "h2"
with torch.no_grad():
    for _ in range(5):
        core.conv_rows(xr, pc, relu=True)
torch.cuda.synchronize()
