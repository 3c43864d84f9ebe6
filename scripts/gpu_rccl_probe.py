"""Can two ranks share ONE GPU under RCCL on this box?  (decides how much of the RCCL path a 1-GPU lease can test)"""
import os, sys, socket
import torch, torch.distributed as dist, torch.multiprocessing as mp


def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "all_reduce ->", t.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank", rank, "FAILED:", type(e).__name__, str(e)[:300], flush=True)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(w, args=(world, port), nprocs=world, join=True)
