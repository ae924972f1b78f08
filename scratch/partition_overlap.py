"""Single-GPU look-ahead (timing experiment): the NEXT batch's tile partition (it depends on the ids only) issued on a second
stream while this step's owner passes run — the item pass's second round leaves most wave slots free.
python scratch/partition_overlap.py [B]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
d = 64
gen = torch.Generator(device=dev).manual_seed(4321)
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii, generator=gen) == 0
su, sp, sn = TripletSampler(iu[tr], ii[tr], NU, NI, seed=99).stream(2 * B)
batches = [tuple(t[k * B:(k + 1) * B].contiguous() for t in (su, sp, sn)) for k in range(2)]

def make():
    return BPRMFStep(torch.randn(NU, d, device=dev) * 0.05, torch.randn(NI, d, device=dev) * 0.05, lr=1e-4, impl="pull")

def plain(step, n):
    for k in range(n): step.step(*batches[k % 2])

def lookahead(step, n, side, ready):
    main = torch.cuda.current_stream()
    step._build_index(*batches[0], 0)
    for k in range(n):
        slot = k % 2
        step._indexed = (slot, batches[slot])              # this batch's index is in its slot
        step.step(*batches[slot])                          # user pass + item pass (the index is found ready)
        done = torch.cuda.Event(); done.record(main)       # (slot 1 - slot was last read by step k - 1: long over)
        with torch.cuda.stream(side):
            step._build_index(*batches[1 - slot], 1 - slot)
            ready[1 - slot].record(side)
        main.wait_event(ready[1 - slot])

for name, fn in (("one stream", plain), ("next partition on a second stream", None)):
    step = make()
    side = torch.cuda.Stream(); ready = [torch.cuda.Event(), torch.cuda.Event()]
    run = (lambda n: plain(step, n)) if fn else (lambda n: lookahead(step, n, side, ready))
    run(20); torch.cuda.synchronize(); t = time.perf_counter()
    run(100); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 100
    step.check()
    print(f"B={B} {name}: {dt * 1e6:.1f} us per step", flush=True)
