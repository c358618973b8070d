"""The end-to-end variant (Inception-shaped trunk on stock MIOpen + bi-LSTM + the hot path) with and without MIOpen's exhaustive kernel search
(torch.backends.cudnn.benchmark): how much of the third-party trunk's 17 ms is the library's default kernel choice?  (development tool)"""
import os, sys, time, importlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
HF = importlib.import_module("attention-gan_amd.backend.functional")
RNN = importlib.import_module("attention-gan_amd.networks.rnn_encoder")
DEV = torch.device("cuda:0")
B = 24
words, sent, lens, reals = bench.synthetic_batch(DEV, B, 1)
g = torch.Generator().manual_seed(3)
for find in (False, True):
    torch.backends.cudnn.benchmark = find
    step = bench.build(DEV, B, HF, "inception")
    rnn = RNN.RNNEncoder(vocabsize=1000, nhidden=bench.EMB).to(DEV).eval(); rnn.freeze_all_weights()
    caps = torch.randint(1, 1000, (B, bench.T), generator=g).to(DEV)
    def e2e():
        with torch.no_grad():
            w_e, s_e = rnn(caps, [bench.T] * B)
        return step.step(w_e.contiguous(), s_e.contiguous(), lens, None, reals)
    for _ in range(6): e2e()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): e2e()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"cudnn.benchmark={find}: {dt * 1e3:.2f} ms per step, {B / dt:.1f} img/s", flush=True)
    del step, rnn
    torch.cuda.empty_cache()
