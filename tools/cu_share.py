"""Do two h2 launch sequences share the chip when each is given a CU budget (iron_set_cu_limit)?  Main stream: get_all on 300k points
under a budget; side stream: 17 dependent get_all launches on 2016 points (the silhouette walk's shape).  Prints the side sequence's
elapsed time beside the main launch's."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from iron_amd import scenes, _lib

dev = torch.device("cuda", 0)
net = scenes.build_networks("S0")["sdf_network"].to(dev)
lib = _lib.load()
big = torch.rand(300000, 3, device=dev) - 0.5
small = torch.rand(2016, 3, device=dev) - 0.5
side = torch.cuda.Stream(device=dev)
net.get_all(big, is_training=False); net.get_all(small, is_training=False)
torch.cuda.synchronize()


def run(main_limit, side_limit, overlap=True):
    m0, m1, s0, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    torch.cuda.synchronize()
    lib.iron_set_cu_limit(main_limit)
    m0.record()
    net.get_all(big, is_training=False)
    m1.record()
    lib.iron_set_cu_limit(side_limit)
    if not overlap:
        torch.cuda.synchronize()
    with torch.cuda.stream(side):
        s0.record(side)
        for _ in range(17):
            net.get_all(small, is_training=False)
        s1.record(side)
    lib.iron_set_cu_limit(0)
    torch.cuda.synchronize()
    return m0.elapsed_time(m1), s0.elapsed_time(s1)


print("alone        main %.2f ms  side %.2f ms" % run(0, 0, overlap=False))
for ml, sl in ((0, 0), (240, 64), (224, 64), (192, 64), (160, 96), (128, 128), (192, 32)):
    a = [run(ml, sl) for _ in range(3)]
    print("main<=%-3d side<=%-3d   main %.2f ms  side %.2f ms" % (ml, sl, min(x[0] for x in a), min(x[1] for x in a)))
