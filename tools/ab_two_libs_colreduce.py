"""A/B of two builds of libemogest_hip.so on the column-reduction entry points (same inputs, outputs compared): usage  python tools/ab_two_libs_colreduce.py OLD.so NEW.so"""
import ctypes as C, sys, torch
old, new = C.CDLL(sys.argv[1]), C.CDLL(sys.argv[2])
dev = torch.device("cuda:0")
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator().manual_seed(1)
for (B, HW, Cc) in ((16, 15872, 32), (5, 3968, 64), (7, 992, 128)):
    rows = B * HW
    a = torch.randn(rows, Cc, generator=g).to(dev); b = torch.randn(rows, Cc, generator=g).to(dev)
    mean = torch.randn(Cc, generator=g).to(dev); out = torch.randn(rows, Cc, generator=g).to(dev)
    res = {}
    for name, lib in (("old", old), ("new", new)):
        lib.eg_colreduce_workspace_floats.restype = C.c_int64
        ws = torch.empty(int(lib.eg_colreduce_workspace_floats(Cc)), device=dev)
        o = []
        o0, o1 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        assert lib.eg_colsum(P(a), P(b), P(o0), P(o1), C.c_int64(rows), Cc, P(ws), st) == 0; o += [o0.clone(), o1.clone()]
        assert lib.eg_colsum(P(a), None, P(o0), P(o1), C.c_int64(rows), Cc, P(ws), st) == 0; o += [o0.clone(), o1.clone()]
        s1, s2 = torch.zeros(B, Cc, device=dev), torch.zeros(B, Cc, device=dev)
        if name == "old" or True:
            rc = lib.eg_se_tail_backward_reduce(P(a), P(out), None, P(b), P(mean), P(s1), P(s2), B, HW, Cc, P(ws), st)
            assert rc == 0, rc
            o += [s1.clone(), s2.clone()]
        # BatchNorm backward (mode 2) and forward (modes 0, 3)
        dx, dg, db = torch.zeros(rows, Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        rstd = torch.rand(Cc, generator=g).to(dev) + 0.5 if name == "old" else res["rstd"]
        res["rstd"] = rstd
        gam = torch.ones(Cc, device=dev)
        assert lib.eg_bn_train_backward(P(a), P(b), P(gam), P(mean), P(rstd), P(dx), P(dg), P(db), C.c_int64(rows), Cc, 0, P(ws), st) == 0
        o += [dg.clone(), db.clone(), dx.clone()]
        y, m2, r2 = torch.zeros(rows, Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        assert lib.eg_bn_train_forward(P(a), P(gam), P(mean), P(y), P(m2), P(r2), P(rm), P(rv), C.c_int64(rows), Cc, C.c_float(0.1), C.c_float(1e-5), P(ws), st) == 0
        o += [m2.clone(), r2.clone()]
        torch.cuda.synchronize()
        res[name] = o
    for i, (x, y) in enumerate(zip(res["old"], res["new"])):
        d = float((x - y).abs().max()); s = float(x.abs().max())
        print(f"B={B} HW={HW} C={Cc} output {i}: max|old-new| {d:.3e} (scale {s:.3e}) {'BITWISE' if torch.equal(x, y) else ''}")
