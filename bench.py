#!/usr/bin/env python3
"""Decode-throughput benchmark of the forward pass on MI355X.

One "step" = one pass of the hot path (transformer_batch + greedy arg-max) over one
ragged batch in which every sequence decodes one token.  Workload at N = 1:
TinyLlama-1.1B shape (random-init fp16 model file in the reference's format), 64
concurrent requests, 1024-token prompts prefilled in 512-token chunks, greedy
decoding (BASELINE.json configs[2]).  With --gpus N the request streams shard as N
data-parallel replicas, one process per GPU, each with its own weights copy and
block pool; there is no data-path collective, so scaling is "weak".

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With --tp the N processes hold ONE model instead (BASELINE.json configs[4]: Llama-2-7B
shape, batch 128): heads and MLP columns are split over the ranks and the two partial
sums per layer are all-reduced on the library's stream (RCCL, or with --tp-exchange peer the
library's own kernel over peer-mapped memory); every rank runs the
same 128 requests, `value` counts them once and scaling is "strong".

Rank 0 prints ONE JSON line.  `value` is decode tokens/s over all ranks with the
batch already resident in HBM (the only host traffic inside a step is the 64-entry
integer batch description going up and 64 token ids coming back).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 matrix peak
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"   # per-kernel HBM bytes per launch from rocprofv3 --pmc passes (tools/gpu_evidence.sh)


# --------------------------------------------------------------------------- helpers
def shard_requests(n_requests_per_gpu: int, rank: int, world: int):
    """Data-parallel placement: request r of the node-wide stream goes to replica
    r mod world.  Every replica gets the same number of requests (weak scaling)."""
    total = n_requests_per_gpu * world
    return [r for r in range(total) if r % world == rank]


def algorithmic_bytes_per_decode_step(shape, batch, ctx_lens, kv_bytes, tp_world=1):
    """Bytes one decode step must move at least once ON ONE GPU: every weight once, K/V
    of every cached position once per kv head, the new K/V rows, the token ids.  Under
    tensor parallelism a GPU holds 1/world of the heads and MLP columns and all of the
    classifier."""
    D, F, L, V = shape.dim, shape.hidden_dim // tp_world, shape.n_layers, shape.vocab_size
    QD, KD = shape.n_heads * shape.head_dim // tp_world, shape.n_kv_heads * shape.head_dim // tp_world
    weights = 2 * (L * (2 * D * QD + 2 * D * KD + 3 * D * F) + V * D) + 4 * (2 * L + 1) * D
    kv_row = L * 2 * KD * kv_bytes
    kv_read = sum(c + 1 for c in ctx_lens) * kv_row
    kv_write = batch * kv_row
    return weights + kv_read + kv_write + batch * (D * 2 + 8)


def prefill_roofline(shape, tokens_per_s, tp_world=1):
    """Prefill is bound by the matrix cores: flops of the projections per token (2 per weight)
    against the dense fp16 peak.  `frac` prices the USEFUL flops; activations enter as fp16
    hi + lo, so the matrix cores issue twice as many (`issued`, `frac_issued`).  Whole-job
    rate over all GPUs that share one model under --tp, per replica otherwise."""
    D, F, L = shape.dim, shape.hidden_dim, shape.n_layers
    QD, KD = shape.n_heads * shape.head_dim, shape.n_kv_heads * shape.head_dim
    flops_per_token = 2.0 * L * (2 * D * QD + 2 * D * KD + 3 * D * F)
    useful = flops_per_token * tokens_per_s / 1e12
    peak = MFMA_PEAK_TFLOPS * tp_world
    return dict(bound="mfma", achieved=round(useful, 1), issued=round(2 * useful, 1), peak=peak,
                unit="TFLOP/s", frac=round(useful / peak, 4), frac_issued=round(2 * useful / peak, 4),
                note="projection flops only; tokens/s includes attention, host scheduling and the decode rows batched with the chunks")


class DistEnv:
    """torch.distributed plumbing: barrier and max/sum over ranks.  Degenerates to
    no-ops for a single process."""

    def __init__(self, want_gpus: int, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.torch = None
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if backend is None:
                # NVLLM_BENCH_BACKEND=gloo: rehearsal of an N-rank run on a box with fewer GPUs than ranks
                # (the ranks then share devices, see `device_index`; the numbers mean nothing, the plumbing runs)
                backend = os.environ.get("NVLLM_BENCH_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.device = torch.device("cuda", self.local_rank) if backend == "nccl" else torch.device("cpu")
        if want_gpus != self.world and self.rank == 0:
            print(f"bench: --gpus {want_gpus} but WORLD_SIZE is {self.world}; using {self.world}", file=sys.stderr)

    def device_index(self) -> int:
        """The GPU of this rank: LOCAL_RANK, wrapped when a rehearsal runs more ranks than the box has GPUs."""
        if self.world == 1:
            return self.local_rank
        n = self.torch.cuda.device_count()   # (does not initialise the GPU)
        return self.local_rank % n if n > 0 else self.local_rank

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def reduce(self, value: float, op: str) -> float:
        if not self.dist:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def broadcast_bytes(self, payload: bytes | None, n: int) -> bytes:
        """`n` bytes from rank 0 to every rank."""
        if not self.dist:
            return payload
        t = self.torch.zeros(n, dtype=self.torch.uint8, device=self.device)
        if self.rank == 0:
            t.copy_(self.torch.frombuffer(bytearray(payload), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())

    def allgather_bytes(self, payload: bytes, n: int) -> bytes:
        """Every rank's `n` bytes, concatenated in rank order, on every rank."""
        if not self.dist:
            return payload
        mine = self.torch.frombuffer(bytearray(payload), dtype=self.torch.uint8).to(self.device)
        parts = [self.torch.zeros(n, dtype=self.torch.uint8, device=self.device) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return b"".join(bytes(t.cpu().numpy().tobytes()) for t in parts)

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def timed_steps(env: DistEnv, sync, step, warmup: int, steps: int):
    """W untimed steps, then exactly K steps between barrier+sync brackets; returns
    the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    sync()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    env.barrier()
    return env.reduce(elapsed, "MAX")


# ------------------------------------------------------------------------ the bench
def ensure_model(path, shape, seed, env):
    from nvllm_amd import synth
    want = 28 + 2 * shape.n_params(shared=False)
    if env.rank == 0 and not (os.path.exists(path) and os.path.getsize(path) == want):
        t0 = time.time()
        big = want > 4e9
        synth.write_model(path, shape, seed=seed,
                          progress=(lambda name: print(f"bench: wrote {name} ({time.time() - t0:.0f}s)", file=sys.stderr, flush=True)) if big else None)
        print(f"bench: wrote {path} ({want / 1e9:.2f} GB) in {time.time() - t0:.1f}s", file=sys.stderr)
    env.barrier()


def run_gpu(args, env, shape, model_path, kv_bytes=None, logits_host=None, steps=None, profile_steps=None):
    """One measured run.  The keyword overrides serve the `variants` block (same workload with the
    reference's fp32 pool rows, or with the logits rows copied to the host every step)."""
    from nvllm_amd import capi, synth
    import copy
    args = copy.copy(args)
    if kv_bytes is not None:
        args.kv_bytes = kv_bytes
    if logits_host is not None:
        args.logits_host = logits_host
    if steps is not None:
        args.steps = steps
    if profile_steps is not None:
        args.profile_steps = profile_steps
    lib = capi.load_hip()           # raises if the HIP library is not built: no fall-back
    lib.nvllm_set_device(env.device_index())
    lib.nvllm_set_kv_elem_bytes(args.kv_bytes)
    lib.nvllm_set_logits_mode(0 if args.logits_host else 1)   # default: only token ids return to the host
    if args.tp_slice and not getattr(run_gpu, "_tp_up", False):
        # ONE rank's share of a model split over `world` ranks, alone on this GPU: the exchange is a callback
        # that does nothing (partial sums stay partial -- the arithmetic and the bytes of the step are those
        # of the rank, its results mean nothing), so the time is the rank's compute without any exchange
        run_gpu._tp_up = True
        rank, world = args.tp_slice
        run_gpu._noop = capi.ALLREDUCE_FN(lambda ctx, buf, count, stream: None)
        if lib.nvllm_tp_init_custom(rank, world, run_gpu._noop, None) != 0:
            raise SystemExit("bench: tensor-parallel slice setup failed")
    elif args.tp and args.tp_exchange == "peer" and not getattr(run_gpu, "_tp_up", False):
        run_gpu._tp_up = True
        # the library's own all-reduce: every rank exports one exchange segment (hipIpc handle) and maps the
        # others'; its kernels store into the peers' HBM over xGMI (no RCCL)
        buf = (C.c_ubyte * capi.TP_PEER_HANDLE_BYTES)()
        if lib.nvllm_tp_peer_handle(buf) != 0:
            raise SystemExit("bench: the peer exchange segment could not be exported")
        handles = env.allgather_bytes(bytes(buf), capi.TP_PEER_HANDLE_BYTES)
        if lib.nvllm_tp_init_peer(env.rank, env.world, handles) != 0:
            raise SystemExit("bench: tensor-parallel setup (peer exchange) failed")
    elif args.tp and not getattr(run_gpu, "_tp_up", False):
        run_gpu._tp_up = True
        # one RCCL communicator over all ranks, owned by the library: its all-reduces run on the
        # library's stream.  (With one rank the same code path runs, the sums have one term.)
        ident = None
        if env.rank == 0:
            buf = (C.c_char * capi.TP_ID_BYTES)()
            if lib.nvllm_tp_unique_id(buf) != 0:
                raise SystemExit("bench: RCCL is not available")
            ident = bytes(buf.raw)
        ident = env.broadcast_bytes(ident, capi.TP_ID_BYTES)
        if lib.nvllm_tp_init(env.rank, env.world, ident) != 0:
            raise SystemExit("bench: tensor-parallel setup failed")
    B = args.batch
    # sequences that finish their prefill early already decode while the others are
    # still prefilling (continuous batching), so budget those steps too
    chunks_per_seq = -(-args.prompt_len // args.chunk)
    per_step = max(1, args.max_seq_len // min(args.chunk, args.prompt_len))
    prefill_steps = -(-B * chunks_per_seq // per_step) + 2
    n_new = prefill_steps + args.warmup + args.steps + args.profile_steps + 8
    if args.prompt_len + n_new > args.max_seq_len:
        raise SystemExit(f"prompt {args.prompt_len} + {n_new} new tokens exceeds max_seq_len {args.max_seq_len}")
    if args.tp:      # every rank runs the same requests
        ids = list(range(B))
        prompts = synth.make_requests(B, args.prompt_len, args.prompt_len, shape.vocab_size, seed=11)
    else:
        ids = shard_requests(B, env.rank, env.world)
        prompts = synth.make_requests(B * env.world, args.prompt_len, args.prompt_len, shape.vocab_size, seed=11)
    eng = capi.Engine(lib, model_path, B, chunk_size=args.chunk, temperature=0.0,
                      max_seq_len=args.max_seq_len, max_steps=1 << 30)
    for slot, r in enumerate(ids):
        eng.set_request(slot, prompts[r], n_new, 0)

    # ---- prefill (chunked), reported alongside
    lib.device_sync()
    t0 = time.perf_counter()
    while True:
        states = [eng.seq_state(i) for i in range(B)]
        if all(s["pos"] >= s["n_prompt"] for s in states):
            break
        eng.step()
    lib.device_sync()
    prefill_s = time.perf_counter() - t0
    st = eng.stats()
    prefill_tokens = st.tokens_forwarded
    ctx0 = [eng.seq_state(i)["pos"] for i in range(B)]

    # ---- timed decode
    eng.reset_stats()
    elapsed = timed_steps(env, lib.device_sync, eng.step, args.warmup, args.steps)
    st = eng.stats()
    assert st.decode_step_tokens == (args.warmup + args.steps) * B, "a timed step was not a pure decode step"
    ctx_mid = [c + args.warmup + args.steps // 2 for c in ctx0]
    step_bytes = algorithmic_bytes_per_decode_step(shape, B, ctx_mid, args.kv_bytes,
                                                   args.tp_slice[1] if args.tp_slice else (env.world if args.tp else 1))

    # ---- per-kernel durations from HIP events on the library stream (same workload,
    #      the steps right after the timed region; events would perturb the timing)
    #      Under --tp every rank takes these steps (a rank cannot all-reduce alone); rank 0's events are reported.
    prof = {}
    if (env.rank == 0 or args.tp) and args.profile_steps > 0:
        lib.nvllm_prof_reset()
        lib.nvllm_prof_enable(1)
        for _ in range(args.profile_steps):
            eng.step()
        lib.nvllm_prof_enable(0)
        for c in range(lib.nvllm_prof_num_classes()):
            n, ms, by, fl = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
            lib.nvllm_prof_get(c, C.byref(n), C.byref(ms), C.byref(by), C.byref(fl))
            if n.value:
                prof[lib.nvllm_prof_name(c).decode()] = dict(launches=n.value, total_ms=ms.value,
                                                             bytes=by.value, flops=fl.value)
    hist = eng.history(0)[-4:].tolist()
    forward_ms = st.decode_step_seconds / max(1, args.warmup + args.steps) * 1e3   # host time spent enqueueing decode passes and waiting for their results (no sampling)
    eng.close()
    return dict(elapsed=elapsed, prefill_s=prefill_s, prefill_tokens=prefill_tokens, step_bytes=step_bytes,
                prof=prof, ctx=int(np.mean(ctx_mid)), tail=hist, forward_ms=forward_ms)


def cpu_baseline(args, shape, model_path, context):
    """The reference CPU path (oracle/_ref, the reference's own kernels compiled here)
    or, where that build is absent, this repository's restatement of it, timed on a
    bounded sample of the same workload: `cpu_steps` decode steps of the same batch at the
    GPU run's context.  The CPU does not prefill (tens of thousands of tokens at tens of
    tokens per second): every sequence starts at position `context` with its blocks
    allocated and a zero-filled fp32 pool behind them -- the arithmetic of a decode step
    does not depend on the values, its results are not compared with anything."""
    from nvllm_amd import capi, synth
    from tests import checkers          # the checker is only ever the thing timed here, as the CPU baseline
    kind = "reference" if os.path.exists(checkers.REF_LIB) else "port"
    lib = checkers.load_ref() if kind == "reference" else checkers.load_oracle()
    B = args.batch
    context = max(1, min(context, args.max_seq_len - args.cpu_steps - 2))
    prompts = synth.make_requests(B, context + 1, context + 1, shape.vocab_size, seed=11)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    sys.stdout.flush()
    os.dup2(devnull, 1)                     # the reference's loader prints a page of text
    try:
        eng = capi.Engine(lib, model_path, B, chunk_size=args.chunk, temperature=0.0,
                          max_seq_len=context + args.cpu_steps + 16, max_steps=1 << 30)
        for i, p in enumerate(prompts):
            eng.set_request(i, p, args.cpu_steps + 1, 0)
            eng.skip_prefill(i, context)
        eng.step()                          # first decode step untimed: it touches the weights
        eng.reset_stats()
        t0 = time.perf_counter()
        for _ in range(args.cpu_steps):
            eng.step()
        dt = time.perf_counter() - t0
        eng.close()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    return dict(value=round(B * args.cpu_steps / dt, 3), unit="tokens/s", cores=cores, kind=kind,
                context_tokens=[context + 1, context + 1 + args.cpu_steps], batch=B, kv_elem_bytes=4,
                sample=f"{args.cpu_steps} decode steps of {B} sequences at context {context + 1}-{context + 1 + args.cpu_steps} "
                       f"({B * args.cpu_steps} tokens, {dt:.1f} s wall, OpenMP; no CPU prefill: blocks allocated, zero-filled fp32 pool)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--tp", action="store_true",
                    help="the ranks hold ONE model, split by heads / MLP columns, with RCCL all-reduces (default workload: "
                         "llama2-7b shape, batch 128, 128-token prompts)")
    ap.add_argument("--tp-exchange", choices=("rccl", "peer"), default=os.environ.get("NVLLM_TP_EXCHANGE", "rccl"),
                    help="with --tp: ncclAllReduce (default), or the library's all-reduce over peer-mapped exchange segments")
    ap.add_argument("--tp-slice", default=None, metavar="R/W",
                    help="with --tp on ONE GPU: time rank R's share of a model split over W ranks with a no-op exchange "
                         "(what a rank of a W-GPU run computes between its all-reduces; results are meaningless)")
    ap.add_argument("--model", default=None)
    ap.add_argument("--batch", type=int, default=None, help="concurrent sequences per GPU (per model under --tp)")
    ap.add_argument("--prompt-len", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=512)
    ap.add_argument("--max-seq-len", type=int, default=None)
    ap.add_argument("--kv-bytes", type=int, default=2, choices=(2, 4),
                    help="KV pool element size: 2 = fp16 pages (default; parity covered by the fp16-pool GPU tests), 4 = the reference's fp32 rows")
    ap.add_argument("--logits-host", action="store_true",
                    help="copy the full fp32 logits rows to the host every step, as the reference's contract has it "
                         "(the PCIe-inclusive variant; not the headline number)")
    ap.add_argument("--profile-steps", type=int, default=8)
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the `variants` block (N = 1 only: the same workload with the fp32 row pool and with logits rows to the host)")
    ap.add_argument("--cpu-steps", type=int, default=8, help="decode steps of the CPU baseline leg (about 1.7 s each on 16 cores); 0 disables it")
    ap.add_argument("--model-dir", default=os.environ.get("NVLLM_MODEL_DIR", "/tmp"))
    args = ap.parse_args()
    if args.tp_slice:
        if not args.tp or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("bench: --tp-slice goes with --tp in a single process")
        args.tp_slice = tuple(int(v) for v in args.tp_slice.split("/"))
    defaults = (("llama2-7b", 128, 128, 512) if args.tp else ("tinyllama-1.1b", 64, 1024, 2048))
    for name, val in zip(("model", "batch", "prompt_len", "max_seq_len"), defaults):
        if getattr(args, name) is None:
            setattr(args, name, val)

    # stdout carries exactly ONE line (the JSON below): whatever libraries print on the way
    # (RCCL's version banner, loader messages) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from nvllm_amd import synth
    env = DistEnv(args.gpus)
    shape = synth.ModelShape.named(args.model, max_seq_len=args.max_seq_len)
    model_path = os.path.join(args.model_dir, f"nvllm_{args.model}_s{args.max_seq_len}_seed0.bin")
    ensure_model(model_path, shape, 0, env)

    r = run_gpu(args, env, shape, model_path)
    N, K, B = env.world, args.steps, args.batch
    variants = None
    if env.world == 1 and not args.tp and not args.no_variants and not args.logits_host and args.kv_bytes == 2:
        # the library's drop-in defaults, on the same workload: the reference's fp32 pool rows, and the
        # reference's result contract (full logits rows in host memory, src/main.c:377; sample() then scans
        # the host row).  Shorter timed regions; never the headline value.
        variants = {}
        vk = max(8, min(K, 16))
        for name, kw in (("fp32_kv_pool", dict(kv_bytes=4)), ("logits_rows_to_host", dict(logits_host=True)),
                         ("fp32_kv_pool_and_logits_rows_to_host", dict(kv_bytes=4, logits_host=True))):
            v = run_gpu(args, env, shape, model_path, steps=vk, profile_steps=0, **kw)
            variants[name] = dict(value=round(B * vk / v["elapsed"], 1), unit="tokens/s", ms_per_step=round(v["elapsed"] / vk * 1e3, 4),
                                  forward_ms_per_step=round(v["forward_ms"], 4), steps=vk, mean_context=v["ctx"],
                                  kv_elem_bytes=kw.get("kv_bytes", 2),
                                  result_handoff="logits rows to host (PCIe-inclusive)" if kw.get("logits_host") else "token ids only")
    ms_per_step = r["elapsed"] / K * 1e3
    value = B * (1 if args.tp else N) * K / r["elapsed"]
    prefill_tps = env.reduce(r["prefill_tokens"] / r["prefill_s"], "MAX" if args.tp else "SUM")

    if env.rank == 0:
        roofline, by_kernel = None, {}
        if r["prof"]:
            # group the timing classes by kernel symbol: all projections are one kernel
            sym = lambda cls: ("k_gemm" if cls.startswith("gemm") else "k_decode_chain" if cls == "decode_chain" else
                               ("k_paged_attn16" if args.kv_bytes == 2 else "k_paged_attn") if cls == "attn" else
                               {"attn_combine": "k_attn_combine", "argmax": "k_argmax", "embed": "k_embed_prescale"}.get(cls, cls))
            for cls, d in r["prof"].items():
                g = by_kernel.setdefault(sym(cls), dict(launches=0, total_ms=0.0, bytes=0.0))
                for k in g:
                    g[k] += d[k]
            pmc = {}
            pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_TRAFFIC_FILE)
            if os.path.exists(pmc_path) and args.model == "tinyllama-1.1b" and args.batch == 64 and args.kv_bytes == 2 and not args.tp:
                pmc = json.load(open(pmc_path))["kernels"]     # measured on this workload in separate --pmc passes
            def roof(name):
                d = by_kernel[name]
                avg_ms = d["total_ms"] / d["launches"]
                achieved = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
                return dict(bound="hbm", kernel=name, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(achieved / HBM_PEAK_GBS, 4),
                            traffic=pmc.get(name, {}).get("hbm_bytes_per_launch"),
                            # PMC counters need their own rocprofv3 passes: the figure is replayed from the
                            # tracked file that tools/gpu_evidence.sh wrote for this workload, not measured in this run
                            traffic_source=(f"profiles/{PMC_TRAFFIC_FILE} (replayed)" if name in pmc else None),
                            avg_kernel_us=round(avg_ms * 1e3, 2),   # dispatch begin -> end, HIP events of the launch
                            launches=d["launches"],
                            algorithmic_bytes_per_launch=int(d["bytes"] / d["launches"]),
                            share_of_step=round(d["total_ms"] / args.profile_steps / ms_per_step, 3))
            roofline = roof(max(by_kernel, key=lambda k: by_kernel[k]["total_ms"]))   # the kernel the step spends most time in
            by_kernel = {k: roof(k) for k in by_kernel}
        cpu = None
        if N == 1 and args.cpu_steps > 0 and not args.tp:
            cpu = cpu_baseline(args, shape, model_path, r["ctx"])
        step_gbs = r["step_bytes"] / (ms_per_step * 1e-3) / 1e9
        line = {
            "metric": (f"decode tokens/sec of rank {args.tp_slice[0]}'s share alone, exchange excluded, Llama-2-7B-shape TP={args.tp_slice[1]} batch {B}"
                       if args.tp_slice else
                       f"decode tokens/sec (node) + % HBM peak, Llama-2-7B-shape TP={N} batch {B}" if args.tp else
                       "decode tokens/sec (node) + % HBM peak, TinyLlama-1.1B paged batch 64"),
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": N, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if args.tp else "weak",
            "vs_baseline": None, "dtype": "f32 (fp16 weights as hi/lo-split fp16 MFMA, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"{args.model} shape, random-init fp16 model.bin, {B} concurrent requests per GPU, "
                                   f"{args.prompt_len}-token prompts prefilled in {args.chunk}-token chunks, greedy decode, "
                                   f"paged KV (block 16, {'fp32' if args.kv_bytes == 4 else 'fp16'} pool), "
                                   f"mean context {r['ctx']} in the timed region",
                       "batch_per_gpu": B, "parallelism": f"one rank of tp{args.tp_slice[1]}, no exchange" if args.tp_slice else (f"tp{N} ({args.tp_exchange} exchange)" if N > 1 else f"tp{N}") if args.tp else (f"dp{N}" if N > 1 else "single"), "kv_elem_bytes": args.kv_bytes,
                       "result_handoff": "logits rows to host (PCIe-inclusive)" if args.logits_host else "token ids only"},
            "roofline": roofline,
            "roofline_by_kernel": by_kernel,
            "cpu_baseline": cpu,
            "variants": variants,
            "step_hbm": {"algorithmic_bytes_per_step": int(r["step_bytes"]), "achieved_GBps": round(step_gbs, 1),
                         "frac_of_peak": round(step_gbs / HBM_PEAK_GBS, 4)},
            "forward_ms_per_step": round(r["forward_ms"], 4),
            "prefill_tokens_per_s": round(prefill_tps, 1),
            "prefill_roofline": None if args.tp_slice else prefill_roofline(shape, prefill_tps, env.world if args.tp else 1),
            "kernels": {k: {"launches": v["launches"], "avg_us": round(v["total_ms"] / v["launches"] * 1e3, 2),
                            "GBps": round(v["bytes"] / max(v["total_ms"], 1e-9) / 1e6, 1)} for k, v in r["prof"].items()},
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if args.tp:
        from nvllm_amd import capi
        capi.load_hip().nvllm_tp_finalize()
    env.close()


if __name__ == "__main__":
    main()
