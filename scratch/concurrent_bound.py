"""Upper bound of what running the two owner passes CONCURRENTLY could buy (timing only: the item pass reads the
coefficients of the previous step, so the numbers are wrong on purpose): user pass on one stream, item pass on another,
joined by events, against the serial step.  python scratch/concurrent_bound.py [B ...]"""
import sys, time, torch
sys.path.insert(0, '.')
from yelprecommendation_amd import engine, _lib
from yelprecommendation_amd.bpr_step import BPRMFStep
from yelprecommendation_amd.data.synthetic import YELP2018_ITEMS as NI, YELP2018_USERS as NU, make_interactions_torch
from yelprecommendation_amd.data.triplets import TripletSampler, split_train_rows
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev).manual_seed(4321)
iu, ii = make_interactions_torch(NU, NI, 47.0, seed=1234, device=dev)
tr = split_train_rows(iu, ii, generator=gen) == 0
lib = _lib.load()
side = torch.cuda.Stream()
for B in [int(a) for a in sys.argv[1:]] or [32768, 65536, 131072, 262144]:
    u, p, n = (t.contiguous() for t in TripletSampler(iu[tr], ii[tr], NU, NI, seed=99).stream(B))
    st = BPRMFStep(torch.randn(NU, 64, device=dev) * 0.05, torch.randn(NI, 64, device=dev) * 0.05, lr=1e-4, impl="pull")
    for _ in range(10): st.step(u, p, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): st.step(u, p, n)
    torch.cuda.synchronize(); serial = (time.perf_counter() - t0) / 200
    ws = st._ws_slots[0]
    main = torch.cuda.current_stream()

    def apply(phases, stream):
        step_size, bc2 = engine.adam_scalars(st.t, st.lr, 0.9, 0.999)
        rc = lib.yr_bpr_mf_pull_apply_ordered(
            st.U.data_ptr(), st._U_alt.data_ptr(), st._pI, st._pmU, st._pvU, st._pmI, st._pvI, None, B, 64, NU, NI,
            1.0 / B, st.lr, step_size, bc2, 0.9, 0.999, 1e-8, 0.0, engine.OPT_ADAM, 0, ws.data_ptr(), ws.numel(),
            st._ppartials, None, None, phases, 0, NI, None, stream.cuda_stream)
        assert rc == 0

    def concurrent():
        st._build_index(u, p, n, 0)
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        apply(engine.PULL_USER_PHASE, main)
        apply(engine.PULL_ITEM_PHASE, side)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)
    for _ in range(10): concurrent()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): concurrent()
    torch.cuda.synchronize(); conc = (time.perf_counter() - t0) / 200
    print(f"B={B}: serial step {serial*1e6:.1f} us, both passes concurrently (two streams, timing only) {conc*1e6:.1f} us", flush=True)
