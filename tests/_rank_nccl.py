"""torch.distributed with the `nccl` backend (= RCCL on ROCm) on this box's GPU(s): every rank binds its device, forms the
process group the bench / driver use for N > 1 and all-reduces a slot buffer on the device.  Started by
semiclassical_amd.distributed.launch_local_ranks with as many ranks as the box has GPUs (one here).

    python tests/_rank_nccl.py OUT.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(rank)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    slots = torch.full((32, 5), float(rank + 1), dtype=torch.float64, device=f"cuda:{rank}")
    buf = slots[:, :4].contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2
    ok = bool((buf == want).all())
    if rank == 0:
        with open(sys.argv[1], "w") as f:
            json.dump({"backend": dist.get_backend(), "world": world, "ok": ok, "sum": float(buf[0, 0])}, f)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
