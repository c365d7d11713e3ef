"""How many weight-gradient GEMMs each backward stage of the N > 1 sequence groups (forced 1-rank RCCL group, launch by launch)."""
import os, sys
os.environ.update(NACF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import nacf_amd
from nacf_amd import synthetic as S
from nacf_amd.runtime import ops
from nacf_amd.runtime.ddp import DataParallel
import bench
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
ops.set_gemm_mode("bf16x3")
opt = bench.make_opt(nacf_amd, "NACF", 20, 10547, sync_bn=True)
model = bench.build_model(nacf_amd, opt, dev)
model.train()
ddp = DataParallel(model, force_collectives=True)
print("split", ddp.bucket_split(), "head", ddp.head_split(), "n", model.flat.grad.numel())
orig_flush = ops.DW_GROUP.flush
def flush():
    n, act = len(ops.DW_GROUP.keep), ops.DW_GROUP.active
    orig_flush()
    if act:
        print("  dw_group flush: %d GEMMs queued" % n)
ops.DW_GROUP.flush = flush
b = S.synth_batch(opt, 128, 60, seed=2)
batch = bench.to_batch(b, dev, True)
from nacf_amd.misc.crit import get_criterion
from nacf_amd.misc.optim import get_optimizer
from nacf_amd.misc.run import get_forword_results
from nacf_amd.runtime.engine import TrainStep
crit, optim = get_criterion(model.opt), get_optimizer(model.opt, model)
engine = TrainStep(model, crit, optim, lambda bb: get_forword_results(model.opt, model, bb, dev), ddp=ddp, graph="off")
for i in range(2):
    print("step", i, "staged", engine.staged, "three", engine.three)
    engine(batch) if i == 0 else engine()
torch.cuda.synchronize()
