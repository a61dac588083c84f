import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l or "librccl" in l))
if order == "torch_first":
    import torch
    print("torch sees", torch.cuda.device_count(), "gpu(s)")
    x = torch.ones(4, device="cuda"); print(float(x.sum()))
    from illuminant_amd import native
    print("ilm sees", native.device_count())
    c = native.Context(0); c.sync(); print("ctx ok")
    uid = native.Group.unique_id(); g = native.Group.rank(0, 0, 1, uid); print("group ok", g.comm_ranks()); g.close()
else:
    from illuminant_amd import native
    print("ilm sees", native.device_count())
    c = native.Context(0); c.sync(); print("ctx ok")
    import torch
    print("maps before torch cuda", maps())
    try:
        print("torch sees", torch.cuda.device_count(), "gpu(s)")
        x = torch.ones(4, device="cuda"); print(float(x.sum()))
    except Exception as e:
        print("torch failed:", e)
print("maps", maps())
