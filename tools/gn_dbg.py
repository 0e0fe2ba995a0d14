import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from mobi_amd import ops, build, _lib
build.build(verbose=False)
lib = _lib.load()
torch.manual_seed(0)
n, hw, c = 2, 4096, 320
x = (torch.randn(n, 1, hw, c) * 2 + 0.7).half().cuda()
g = (torch.randn(c) * 0.2 + 1).cuda(); b = (torch.randn(c) * 0.2).cuda()
os.environ["MOBI_GN_COOP"] = "1"; lib.mobi_tuning_reload()
out = torch.empty_like(x)
ws = torch.zeros(lib.mobi_groupnorm_workspace_bytes(n, hw) // 4, device="cuda", dtype=torch.float32)
sync = torch.zeros(64, device="cuda", dtype=torch.int32)
p = _lib.GroupNormParams()
p.src0, p.c0, p.c1, p.batch, p.hw = x.data_ptr(), c, 0, n, hw
p.gamma, p.beta, p.eps, p.silu = g.data_ptr(), b.data_ptr(), 1e-5, 0
p.out, p.ws, p.dtype, p.sync = out.data_ptr(), ws.data_ptr(), 0, sync.data_ptr()
print(lib.mobi_groupnorm(C.byref(p), None)); torch.cuda.synchronize()
chunks = 32
part = ws[: n * chunks * 64].view(n, chunks, 32, 2).cpu()
xf = x.float().cpu().view(n, chunks, hw // chunks, 32, c // 32)
rs = xf.sum(dim=(2, 4)); rq = (xf * xf).sum(dim=(2, 4))
print("sum  part vs ref", part[0, :3, :4, 0], rs[0, :3, :4])
print("sq   part vs ref", part[0, :3, :4, 1], rq[0, :3, :4])
print("ratio sq", (part[..., 1] / rq)[0, :2, :8])
print("sync", sync[:4])
