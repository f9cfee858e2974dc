"""Step time after start / idle / spin-up: shows how long the device needs under load before the same step
runs at its steady speed (see bench.py spin_up).  python tools/ramp_probe.py"""
import importlib, pathlib, sys, time, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench
mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
batches, _ = bench.make_batches(8, 8192, seed=1000, device=dev)
tr = bench.Trainer(mf, dev, "adam", 0)
reserved, host_ms = [], []
def run(n, tag):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        h0 = time.perf_counter()
        ev[i].record(); tr.step(batches[i % 8]); reserved.append(torch.cuda.memory_reserved() >> 20)
        host_ms.append(1e3 * (time.perf_counter() - h0))
    ev[n].record(); torch.cuda.synchronize()
    print(tag, " ".join(f"{ev[i].elapsed_time(ev[i+1]):.2f}" for i in range(n)))
if len(sys.argv) > 1 and sys.argv[1] == "lossspin":      # spin with the sweeps themselves, on scratch data
    g = torch.Generator(device="cpu").manual_seed(3)
    su = torch.nn.functional.normalize(torch.randn(8192, 128, generator=g), dim=-1).to(dev).requires_grad_()
    sv = torch.nn.functional.normalize(torch.randn(16384, 128, generator=g), dim=-1).to(dev).requires_grad_()
    fn = mf.losses.InfomationNoiseContrastiveEstimationLoss()
    b0 = batches[0]
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < float(sys.argv[2]):
        for _ in range(50):                      # back to back: the host runs ahead, the device never idles
            fn(su, sv, b0["target"], item_idx=b0["item"], pos_idx=b0["pos"]).backward()
        torch.cuda.synchronize()
        n += 50
    print("loss spin iterations:", n)
if len(sys.argv) > 1 and sys.argv[1] == "spinfirst":
    bench.spin_up(dev, float(sys.argv[2]) if len(sys.argv) > 2 else 0.2)
run(40, "cold      :")
print("host enqueue ms per cold step:", " ".join(f"{x:.2f}" for x in host_ms))
run(12, "continued :")
time.sleep(0.5)
run(20, "after 0.5s idle:")
bench.spin_up(dev, 0.2)
run(12, "after 0.2s GEMM spin:")
time.sleep(0.5); 
x = torch.randn(8192, 128, device=dev); y = torch.randn(16384, 128, device=dev)
t0=time.perf_counter()
while time.perf_counter()-t0 < 0.2:
    for _ in range(20): (x @ y.T).sum()
    torch.cuda.synchronize()
run(12, "after idle + 0.2s small-GEMM spin:")
