"""Which torch (aten) device kernels does one training step still launch, and from which line of the package?
Runs the NACF bench step launch-by-launch under torch.profiler with Python stacks and prints every aten operator that
launched a device kernel, with shapes and the innermost frames of nacf_amd / torch.autograd that called it.
usage (GPU box): python tools/aten_in_step.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import nacf_amd  # noqa: E402
from nacf_amd.runtime import ops  # noqa: E402
from nacf_amd import synthetic as O  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
opt = bench.make_opt(nacf_amd, "NACF", 20, 10547)
model = bench.build_model(nacf_amd, opt, dev)
model.train()
batch = bench.to_batch(O.synth_batch(opt, B, 60, seed=1), dev, True)
engine, crit, optim = bench.make_engine(model, dev, batch, graph="off", eager_steps=3)


def step():
    optim.zero_grad()
    loss_ = crit.get_loss(engine.forward(engine.static))
    with ops.dw_group():
        loss_.backward()
    optim._optimizer.step(grad_scale=1.0)


step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    if any(c.kernels for c in ev.cpu_children if c.name.startswith("aten::")):
        continue                                   # report the innermost operator that owns the kernel
    frames = [f for f in (ev.stack or []) if "nacf" in f or "non-autoregressive" in f or "bench.py" in f][:3]
    key = (ev.name, str(ev.input_shapes), tuple(frames))
    e = seen.setdefault(key, [0, 0.0, [k.name[:70] for k in ev.kernels]])
    e[0] += 1
    e[1] += sum(k.duration for k in ev.kernels)
for (name, shapes, frames), (n, us, kern) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    print("%-22s x%d %7.1f us  %s\n    kernels: %s" % (name, n, us, shapes, kern))
    for f in frames:
        print("    at", f)
    if not frames:
        print("    at (autograd engine: gradient accumulation / no package frame)")
