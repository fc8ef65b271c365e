"""tools/rccl_self_message_probe.py -- one rank over the nccl (= RCCL) backend: all_to_all_single of 0.16 / 0.8 / 1.6 / 2.4 GB to oneself,
checked against the input.  On this image the two largest return at once with other bytes (profiles/r03_rccl_self_message_probe.txt);
radixhashjoin_amd/sharded.py therefore copies a rank's own segment itself and cuts peer segments into messages of <= 512 MiB.
    HSA_ENABLE_IPC_MODE_LEGACY=0 python tools/rccl_self_message_probe.py"""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
for n in (20_000_000, 100_000_000, 200_000_000, 300_000_000):
    a = torch.arange(n, dtype=torch.int64, device=dev); b = torch.empty_like(a)
    torch.cuda.synchronize(); t0 = time.time()
    dist.all_to_all_single(b, a, [n], [n]); torch.cuda.synchronize()
    print(f"all_to_all_single self {n * 8 / 1e9:.2f} GB: {time.time() - t0:.3f} s ok={bool((a == b).all())}", flush=True)
    del a, b
dist.destroy_process_group()
