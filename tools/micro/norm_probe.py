"""Which fp32 operation order does torch.norm(v, dim=-1) use for 3-vectors on this build?  (probe for
nerf_amd_assemble_rays; run on the GPU box)"""
import torch
g = torch.Generator().manual_seed(3)
v = torch.randn(100000, 3, generator=g).cuda()
nt = torch.norm(v, dim=-1, keepdim=True)
x, y, z = [v[:, i:i+1].double() for i in range(3)]
def f32(t): return t.float().double()
xx, yy, zz = f32(x*x), f32(y*y), f32(z*z)
cands = {
 "fma_xyz": f32(z*z + f32(y*y + f32(x*x))),
 "(xx+yy)+zz": f32(f32(xx + yy) + zz),
 "(xx+zz)+yy": f32(f32(xx + zz) + yy),
 "xx+(yy+zz)": f32(xx + f32(yy + zz)),
 "exact": f32(x*x+y*y+z*z),
}
for k, n2 in cands.items():
    n = torch.sqrt(n2).float()           # sqrt in double then round: correctly rounded sqrt of the fp32 value
    n_dev = torch.sqrt(n2.float())       # the device's own fp32 sqrt
    d = (n.view(torch.int32) - nt.view(torch.int32)).abs()
    d2 = (n_dev.view(torch.int32) - nt.view(torch.int32)).abs()
    print("%-12s mismatches (rounded sqrt): %6d max ulp %d | (device sqrt): %6d max ulp %d" % (k, int((d != 0).sum()), int(d.max()), int((d2 != 0).sum()), int(d2.max())))
# sqrt itself
a = torch.rand(100000, device="cuda") * 10
print("device sqrt vs correctly rounded:", int((torch.sqrt(a) != torch.sqrt(a.double()).float()).sum()))
q = v / nt
print("div vs IEEE:", int((q != (v.double() / nt.double()).float()).sum()))
