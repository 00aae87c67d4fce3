#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame on cornell-box 800x600 spp=512
(BASELINE.json configs[1]) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W
      N > 1, started plainly: this process touches no GPU; it starts N ranks of itself (RANK / LOCAL_RANK /
      WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), one process per GPU over RCCL, and
      relays rank 0's JSON line -- the same ranks `python -m torch.distributed.run --nproc-per-node N bench.py
      --gpus N ...` starts (that launcher works too: the ranks find WORLD_SIZE in the environment).
  python bench.py --gpus N --engine multi
      ONE process, N devices: the in-library entry crt_multi_render (include/crt.h) -- a host thread and a stream
      per device, one ncclAllGather over RCCL, no torch.distributed.

A "step" is one complete frame: every rank path-traces its interleaved 8x8 pixel tiles with the HIP kernels of
libcrt.so, the compact RGB8 tile buffers are all-gathered over RCCL and de-interleaved into the final image.
The total work is fixed as N grows ("strong" scaling).  One frame at a time (ms_per_step is a frame's latency);
--frames-in-flight 2 overlaps consecutive frames (an option, see its help text).

A ray is one closest-hit query of the reference (DeviceBVH::intersect): primary, bounce, shadow and specular-probe
rays; the count is deterministic given (scene, config, seed) and comes from the kernel's counters.

roofline -- dominant kernel k_mega3 (persistent path-tracing megakernel, one launch per frame).  The kernel's time is
measured live (HIP events on the launching stream); the counts it is set against come from rocprofv3 PMC passes of
the same workload, kept in profiles/pmc_latest.json and STAMPED with a hash of the kernel sources and build flags:
when the hash differs from the library that is running (or CRT_LIB_PATH loads another library), the counter-based
fields are null (never a stale number).  Three candidates are priced against DATASHEET peaks only (MI355X_MICROARCH.md); each
fraction is reported as measured (`frac_raw`) and clamped to 1 (`frac`), a raw value above 1 is flagged `suspect` and cannot name the
bound; `roofline.bound` is the unit that is busiest BY THE HARDWARE'S OWN COUNTERS (pick_bound):
    memory_side  bytes that cross the L2's fabric side (32 x the TCC_EA0_*_DRAM_32B request counters: exact; or 2 x FETCH_SIZE +
               WRITE_SIZE when only those were collected) / t against the 8 TB/s HBM peak.  NOT HBM bytes: those counters sit in
               front of the Infinity Cache and count what it answers, no counter behind it is exposed, and this kernel's whole
               resident set fits that cache -- the figure is path-state traffic between the XCDs' L2s and the memory side;
    valu_issue SQ_INSTS_VALU / t against SIMDs x clock / 2 (the guide's 2 cycles per wave64 instruction) as `frac`; what enters the
               choice of the bound is `valu_busy_hw` (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles: how busy the vector pipes were, by the
               hardware's own counter).  Beside them: `frac_at_mix_cost` (the count priced with the cycles per instruction of THIS
               kernel's opcode mix: the exact census of tools/bbprof x the per-opcode costs of tools/valu_issue_gen.py) and
               `arith_share` (share of the vector instructions that are box / triangle arithmetic);
    l2         TCC requests x 128 B / t against the L2 peak;
    vmem_ta    the vector-memory pipe: TA_TA_BUSY (texture addressers busy) / all cycles as `frac`; beside it the rate of wave64 vector-memory
               instructions against one per 16 cycles per CU.  roofline.sensitivity (profiles/r06_experiments/sensitivity.json) holds the
               A/B pairs that say which of these the frame time actually follows.
The SURVEY 8(d) contract figure -- algorithmic bytes of the REFERENCE traversal's visit set, B_ray = 64 x inner
visits + 8 x leaf visits + 36 x triangle tests + 16 x hits, counted by the exhaustive kernel on a spp=8 slice --
stays as `contract_*` side fields: the production traversal walks a 4-wide SAH tree over the reference's leaves and
prunes, so it does not perform that work and the figure exceeds the HBM peak (it is not a bound).

parity (same run): the frame the cpu_baseline leg renders on the oracle (800x600 at --cpu-spp) is
rendered on the GPU too and compared over all pixels; the FAST traversal is also compared with the exhaustive
REFERENCE traversal on that slice.

cpu_baseline: the single-threaded CPU oracle (a port of the reference algorithm, oracle/crt_oracle.cpp) timed on
this host on the same scene at 800x600 spp=--cpu-spp, rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# BASELINE.json configs: scene, width, height, spp (C1 is C2's scene at spp 2 on the CPU: the cpu_baseline leg)
WORKLOADS = {"c2": ("cornell-box", 800, 600, 512), "c3": ("veach-mis", 800, 600, 1024),
             "c4": ("cornell-box", 3840, 2160, 256), "c5": ("veach-mis", 1920, 1080, 4096)}
HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E (MI355X_MICROARCH.md)
L2_PEAK_GBPS = 34500.0       # 8 XCDs x 16 channels x 128 B/clk at 2.1 GHz
N_SIMDS = 1024               # 256 CUs x 4
CLOCK_GHZ = 2.4
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")
CENSUS_FILE = os.path.join(ROOT, "profiles", "bbprof_latest.json")
GUIDE_VALU_CYCLES = 2.0      # MI355X_MICROARCH.md: a wave64 fp32 instruction issues in 2 cycles (the peak of the guide's convention)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)  # (the first two frames after start-up run 1 % slower: clocks and caches settle)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="a BASELINE.json configuration: c2 (default) cornell-box 800x600 spp 512, c3 veach-mis 800x600 spp 1024, "
                         "c4 cornell-box 3840x2160 spp 256, c5 veach-mis 1920x1080 spp 4096 (c4 / c5 are the 8-GPU configurations; "
                         "they run on any N).  --scene / --width / --height / --spp override single fields")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--traversal", default="exact", choices=["exact", "fast", "reference"],
                    help="exact (default): provably the reference's frame; fast: + distance pruning (measured rate of lost rays, include/crt.h)")
    ap.add_argument("--engine", default="procs", choices=["procs", "multi"],
                    help="procs: one process per GPU, torch.distributed over RCCL; multi: one process, crt_multi_render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the veach-mis 800x600 spp=1024 sub-record")
    ap.add_argument("--cpu-spp", type=int, default=8)
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2],
                    help="1 (default): one frame at a time, ms_per_step is a frame's latency.  2: frames are pipelined over two device "
                         "replicas of the scene and two streams of different priority, so that the end of one launch (every wave running "
                         "its pool of paths dry) overlaps the start of the next: measured 0.5 %% / 1 %% / 2.5 %% / 4.9 %% on a rank's "
                         "share of C2 at 1 / 2 / 4 / 8 ranks (tools/pipeline_probe.py), but k_accumulate of frame i then waits for "
                         "kernel i+1 (no free VGPRs) and a frame's latency doubles -- an option, not the reported number")
    ap.add_argument("--save-png", default=None)
    ap.add_argument("--no-large-scene", action="store_true", help="skip the large-mesh sub-record (a 126 000-triangle variant of the scene, and the stand-in under the large-mesh layout)")
    a = ap.parse_args()
    w = WORKLOADS[a.workload or "c2"]
    explicit = any(v is not None for v in (a.scene, a.width, a.height, a.spp))
    a.scene = a.scene or w[0]
    a.width = a.width or w[1]
    a.height = a.height or w[2]
    a.spp = a.spp or w[3]
    a.workload_id = next((k for k, v in WORKLOADS.items() if v == (a.scene, a.width, a.height, a.spp)), None) if (explicit or a.workload is None) else a.workload
    return a


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly: start the N ranks as child processes BEFORE anything touches a GPU
    (a process that has initialised HIP must never exec or fork workers) and relay their exit status."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "CRT_BENCH_SPAWNED": "1"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # if one rank dies the others would wait in a collective until its time-out: end them (by their own PIDs) as soon as one fails
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = max(rc, abs(code))
                t_grace = time.time() + 5.0  # the others usually fail for the same reason a moment later: let them say so
                while time.time() < t_grace and any(q.poll() is None for q in live):
                    time.sleep(0.1)
                for q in live:
                    if q.poll() is None:
                        q.terminate()
                t_end = time.time() + 10.0
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                live = []
                break
    return rc


def load_pmc(workload_key):
    """Counters of `workload_key` from profiles/pmc_latest.json if they were collected on the code that is running."""
    from cudaraytracing_amd import build as B
    if os.environ.get("CRT_LIB_PATH"):
        return None, "CRT_LIB_PATH loads a library of unknown sources: no hash-stamped counters are attached"
    try:
        d = json.load(open(PMC_FILE))
    except Exception:
        return None, "profiles/pmc_latest.json missing"
    if d.get("src_hash") != B.source_hash():
        return None, "profiles/pmc_latest.json was collected on other kernel sources (%s, running %s): counter-based fields dropped" % (
            d.get("src_hash"), B.source_hash())
    w = d.get("workloads", {}).get(workload_key)
    if not w:
        return None, "profiles/pmc_latest.json has no workload %s" % workload_key
    return w, None


def load_census(workload_key):
    """The instruction census of `workload_key` (tools/bbprof/census.py) if it was taken on the code that is running."""
    from cudaraytracing_amd import build as B
    if os.environ.get("CRT_LIB_PATH"):
        return None
    try:
        d = json.load(open(CENSUS_FILE))
    except Exception:
        return None
    if d.get("src_hash") != B.source_hash():
        return None
    return d.get("workloads", {}).get(workload_key)


def bounds_from_pmc(pmc, k_s, census=None):
    """The candidate bounds for one launch of duration k_s seconds, each against a datasheet peak (MI355X_MICROARCH.md).  A fraction is
    reported as measured (`frac_raw`) beside the value clamped to 1 (`frac`); a raw value above 1 means a wrong peak, a miscounted
    counter or a time / counter mismatch and is flagged `suspect` -- such a candidate never names the bound.
    census (tools/bbprof): exact dynamic opcode counts priced with measured per-opcode issue costs -- side fields only."""
    out = {}

    def frac(x):
        return round(min(1.0, x), 4)

    def put_frac(d, x, key="frac"):
        d[key] = frac(x)
        d[key + "_raw"] = round(x, 4)
        if x > 1.0:
            d["suspect"] = True
    if pmc.get("FETCH_SIZE") is not None and pmc.get("WRITE_SIZE") is not None:
        # What crosses the L2's fabric side per launch.  NOT HBM bytes: these counters sit in front of the Infinity Cache and count what
        # it answers (MI355X_MICROARCH.md "HBM"); rocprofv3 on this part exposes no counter behind it (rocprofv3 --list-avail, round 5), and
        # this kernel's resident set (scene 3.6 MB + path-state planes 42 MB + the records in use) fits that cache 5 times over.  Priced
        # against the HBM peak all the same -- the only datasheet number for that side of the chip -- and named for what it is.
        traffic = 2.0 * pmc["FETCH_SIZE"] * 1024.0 + pmc["WRITE_SIZE"] * 1024.0
        plain = pmc["FETCH_SIZE"] * 1024.0 + pmc["WRITE_SIZE"] * 1024.0
        exact = None
        if pmc.get("fabric_read_bytes") is not None and pmc.get("fabric_write_bytes") is not None:
            exact = pmc["fabric_read_bytes"] + pmc["fabric_write_bytes"]   # 32-byte units counted as such (TCC_EA0_*_DRAM_32B): no assumed request size
        use = exact if exact is not None else traffic
        m = {"achieved": round(use / k_s / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "bytes_per_launch": use,
             "bytes_source": "32 x (TCC_EA0_RDREQ_DRAM_32B + TCC_EA0_WRREQ_WRITE_DRAM_32B + TCC_EA0_WRREQ_ATOMIC_DRAM_32B)" if exact is not None
                             else "2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction, exact for wide streaming reads only)",
             "by_fetch_write_size": {"bytes_x2": traffic, "bytes_without_x2": plain,
                                     "frac_x2": frac(traffic / k_s / 1e9 / HBM_PEAK_GBPS), "frac_without_x2": frac(plain / k_s / 1e9 / HBM_PEAK_GBPS)},
             "note": "L2 <-> fabric traffic (Infinity-Cache hits included), not DRAM traffic: what the XCDs' L2s exchange with the memory side, "
                     "i.e. path state that does not fit 8 x 4 MB of L2"}
        if exact is not None:
            m["read_bytes"] = pmc["fabric_read_bytes"]; m["write_bytes"] = pmc["fabric_write_bytes"]
            m["read_bytes_per_request"] = pmc.get("fabric_read_bytes_per_request")
        put_frac(m, use / k_s / 1e9 / HBM_PEAK_GBPS)
        out["memory_side"] = m
    if pmc.get("SQ_INSTS_VALU") is not None:
        peak = N_SIMDS * CLOCK_GHZ / GUIDE_VALU_CYCLES
        ach = pmc["SQ_INSTS_VALU"] / k_s / 1e9
        v = {"achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
             "peak_note": "SIMDs x clock / 2 cycles per wave64 instruction (the guide's figure)", "lane_utilization": pmc.get("valu_lane_utilization")}
        put_frac(v, ach / peak)
        if pmc.get("SQ_ACTIVE_INST_VALU") is not None:
            # SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD's vector pipe executed: x 4 / (SIMDs x clock x t)
            put_frac(v, pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * CLOCK_GHZ * 1e9 * k_s), "valu_busy_hw")
        if census:
            cyc = float(census["cycles_per_valu"])
            v["frac_at_mix_cost"] = frac(ach * cyc / (N_SIMDS * CLOCK_GHZ))
            v["mix_cycles_per_instr"] = round(cyc, 4)
            rng = census.get("cycles_per_valu_range")
            if rng:
                v["frac_at_mix_cost_range"] = [frac(ach * rng[0] / (N_SIMDS * CLOCK_GHZ)), frac(ach * rng[1] / (N_SIMDS * CLOCK_GHZ))]
            v["arith_share"] = round(census["arith_share"], 4) if census.get("arith_share") is not None else None
            v["traversal_bookkeeping_share"] = round(census["traversal_bookkeeping_share"], 4) if census.get("traversal_bookkeeping_share") is not None else None
            v["path_logic_share"] = round(census["path_logic_share"], 4) if census.get("path_logic_share") is not None else None
            v["census"] = {"valu_instructions_per_launch": census["valu_instructions_per_launch"], "lane_utilisation": round(census["lane_utilisation"], 4),
                           "tiers": {k: round(x, 4) for k, x in census["tiers"].items()}, "unpriced_share": round(census["unpriced_share"], 5),
                           "pmc_over_census_instructions": round(pmc["SQ_INSTS_VALU"] / census["valu_instructions_per_launch"], 4),
                           "source": "tools/bbprof (exact dynamic opcode counts of the production ISA) x tools/valu_issue_gen.py (measured per-opcode issue costs)"}
        out["valu_issue"] = v
    if pmc.get("ta_busy_frac") is not None:
        # The vector-memory pipe: cycles in which a CU's texture addresser (the unit every global load / store of its 16 waves passes) was busy
        # / all cycles -- the hardware's own counter, like valu_busy_hw.  Round 6's A/B pairs (profiles/r06_experiments/sensitivity.json) say
        # this is the unit the frame time follows: 9 % fewer vector ALU instructions moved nothing, one more divergent load per inner visit
        # costs 2 - 7 %.  `achieved` / `peak` in wave64 vector-memory instructions per second against one per 16 cycles per CU (64 lanes, 4 per clock).
        ach = (pmc.get("SQ_INSTS_VMEM") or 0.0) / k_s / 1e9
        peak = 256 * CLOCK_GHZ / 16.0
        t = {"achieved": round(ach, 2), "peak": round(peak, 2), "unit": "G wave-instr/s (vector memory)", "peak_note": "256 CUs x clock / 16 cycles per wave64 instruction (4 lanes per clock through the addresser)",
             "l1_tag_lookups_per_instr": pmc.get("l1_tag_lookups_per_vmem_instr"), "l2_read_requests_per_instr": pmc.get("l2_read_requests_per_vmem_instr"),
             "l1_miss_latency_cycles": pmc.get("l1_read_latency_cycles")}
        put_frac(t, ach / peak, "frac_at_16_cycles")
        put_frac(t, float(pmc["ta_busy_frac"]))
        out["vmem_ta"] = t
    if pmc.get("TCC_REQ_sum") is not None or (pmc.get("TCC_HIT_sum") is not None and pmc.get("TCC_MISS_sum") is not None):
        req = pmc.get("TCC_REQ_sum")
        if req is None:
            req = pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]
        ach = req * 128.0 / k_s / 1e9
        out["l2"] = {"achieved": round(ach, 1), "peak": L2_PEAK_GBPS, "unit": "GB/s"}
        put_frac(out["l2"], ach / L2_PEAK_GBPS)
    return out


def pick_bound(bounds):
    """The unit that is busiest by the hardware's own account names the bound: for the vector pipes that is `valu_busy_hw` (cycles in
    which a SIMD's vector unit executed / all SIMD cycles -- a counter, not a priced instruction mix), for the memory side and L2 their
    bytes against the datasheet peak, for the vector-memory pipe `vmem_ta` the texture addressers' busy cycles / all cycles (a counter too).
    A candidate whose raw fraction exceeds 1 is suspect and cannot win."""
    best, best_f = None, -1.0
    for name, b in bounds.items():
        if b.get("suspect"):
            continue
        f = b.get("valu_busy_hw", b["frac"]) if name == "valu_issue" else b["frac"]
        if f > best_f:
            best, best_f = name, f
    return best, best_f


def device_identity(torch, device):
    """(63-bit key, what it was made of) of the physical GPU behind `device` on this host: a hash over the host name and the device's
    UUID (torch.cuda.get_device_properties(...).uuid), else over its PCI domain:bus:device; (None, reason) when neither can be read."""
    import hashlib
    props = None
    try:
        props = torch.cuda.get_device_properties(device)
    except Exception as e:
        return None, "get_device_properties failed: %s" % type(e).__name__
    ident, src = None, None
    u = getattr(props, "uuid", None)
    if u is not None and str(u).strip("0-") != "":
        ident, src = "uuid:" + str(u), "host name + device UUID"
    else:
        dom, bus, dev = (getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is not None and int(bus) >= 0:
            ident, src = "pci:%s:%s:%s" % (dom, bus, dev), "host name + PCI domain:bus:device"
    if ident is None:
        return None, "neither a UUID nor a PCI bus id is exposed"
    h = hashlib.sha256((socket.gethostname() + "|" + ident).encode()).digest()
    return int.from_bytes(h[:8], "little") >> 1, src


def main_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    import cudaraytracing_amd as crt
    from cudaraytracing_amd import build as B
    from cudaraytracing_amd.distributed import render_sharded

    multi = args.engine == "multi"
    rank = 0 if multi else int(os.environ.get("RANK", "0"))
    local_rank = 0 if multi else int(os.environ.get("LOCAL_RANK", "0"))
    world = 1 if multi else int(os.environ.get("WORLD_SIZE", "1"))
    if not multi and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # CRT_BENCH_ONE_DEVICE=1 (testing only): all ranks share cuda:0 (procs: gather over gloo; multi: peer copies) to exercise
    # the N > 1 code path on a single-GPU box (RCCL refuses two ranks on one device)
    one_device = os.environ.get("CRT_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    n_dev = torch.cuda.device_count()
    if (args.gpus if multi else local_rank + 1) > n_dev and not one_device:
        raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s) (CRT_BENCH_ONE_DEVICE=1 rehearses the N > 1 path on one)" % (args.gpus, n_dev))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if one_device else "nccl"
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    # What the collective backend really spans (the first run on an 8-GPU node should prove RCCL saw 8 ranks on 8 devices): every rank
    # adds a device-resident 1 through the process group, and the PCI bus ids of the ranks' devices are all-gathered over the same group.
    collective_proof = None
    if world > 1:
        ones = torch.ones(1, dtype=torch.int32, device=(torch.device("cpu") if backend == "gloo" else device))
        dist.all_reduce(ones)
        # the PHYSICAL device of every rank, and nothing else, is the key (a key that holds the local rank counts ranks, not
        # devices): 8 bytes of a hash over (host name, the device's UUID -- or its PCI domain:bus:device where this torch exposes no UUID);
        # a rank that can read neither contributes "unknown" and the count is then reported as None, never as a number
        dev_key, dev_id_src = device_identity(torch, device)
        mine = torch.tensor([dev_key if dev_key is not None else 0, 0 if dev_key is not None else 1], dtype=torch.int64, device=ones.device)
        allb = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)
        unknown = sum(int(t[1]) for t in allb)
        collective_proof = {"backend": backend, "ranks_counted_by_all_reduce": int(ones.item()),
                            "distinct_devices": (len({int(t[0]) for t in allb}) if unknown == 0 else None),
                            "device_identity": dev_id_src, "ranks_with_unknown_device": unknown}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    cfg = os.path.join(ROOT, "scenes", args.scene, "config.json")
    task = crt.Task(cfg, base_dir=ROOT)
    c0 = time.perf_counter()
    scene = crt.Scene.from_task(task, args.width, args.height)
    setup = {"load_and_host_bvh_ms": round((time.perf_counter() - c0) * 1e3, 2)}
    eye = task.eye_pos
    inv_view = crt.get_inverse_view_matrix(task.eye_pos, task.lookat, task.up)
    fov = crt.fov_to_radians(task.fov_y)
    trav = {"fast": crt.TRAVERSAL_FAST, "exact": crt.TRAVERSAL_EXACT, "reference": crt.TRAVERSAL_REFERENCE}[args.traversal]
    mr = None
    if multi:
        devs = [0] * args.gpus if one_device else list(range(args.gpus))
        mr = crt.MultiRender(scene, args.spp, task.P_RR, task.light_sample_n, devices=devs,
                             gather=crt.GATHER_COPY if one_device else (crt.GATHER_RCCL if args.gpus > 1 else crt.GATHER_AUTO))
        mr.seed = args.seed
        mr.traversal = trav
    render = crt.Render(scene, args.spp, task.P_RR, task.light_sample_n, device=local_rank)
    render.seed = args.seed
    render.traversal = trav
    pipelined = args.frames_in_flight > 1 and not multi
    render2 = None
    if pipelined:
        render2 = crt.Render(scene, args.spp, task.P_RR, task.light_sample_n, device=local_rank)
        render2.seed = args.seed
        render2.traversal = trav

    if rank == 0:
        # scene set-up on the device (not part of a step): the reference BVH (byte-identical to the host build above) and the SAH tree of
        # the FAST traversal, both built on the GPU -- crt_host_scene_set_bvh_device / crt_scene_create
        s2 = crt.Scene.from_task(task, args.width, args.height, bvh_device=local_rank)
        s2 = crt.Scene.from_task(task, args.width, args.height, bvh_device=local_rank)  # (second build: code objects loaded)
        same = s2.nodes().tobytes() == scene.nodes().tobytes() and s2.triangles().tobytes() == scene.triangles().tobytes()
        bi = s2.bvh_build_info
        ai = render.accel_info()
        setup.update({"bvh_device": {"total_ms": round(bi["total_ms"], 2), "level_loop_ms": round(bi["device_ms"], 2), "host_sorts": bi["host_sorts"],
                                     "host_triangles": bi["host_triangles"], "byte_identical_to_host_build": bool(same)},
                      "sah_tree": {"on_device": bool(ai["sah_on_device"]), "ms": round(ai["sah_ms"], 2), "device_ms": round(ai["sah_device_ms"], 2),
                                   "leaves": ai["n_leaves"], "nodes4": ai["n_nodes4"]},
                      # the HIP runtime's one-off start in this process (context, code objects), paid by whatever touches the device first:
                      # timed by itself (crt_accel_info.runtime_init_ms); rounds 1-3 booked it on the SAH build ("147 ms")
                      "runtime_init_ms": round(ai["runtime_init_ms"], 2)})
        s2.free()

    def step():
        if multi:
            mr.run_view(eye, inv_view, fov, want_mean=False, width=args.width, height=args.height, to_host=False)
            st = dict(mr.stats)
            st["kernel_ms"] = mr.info["max_kernel_ms"]
            return None, st
        return render_sharded(render, eye, inv_view, fov, args.width, args.height, rank, world, device)

    kernel_ms, logic_ms, kernel_launches = [], [], []
    kernel_ms_missing = 0   # frames without a megakernel launch to time (the fallback pipeline)
    rays_local = 0
    untraced_local = 0
    img = None
    frame_latency_ms = None
    if pipelined:
        from cudaraytracing_amd.distributed import FramePipeline
        # one unpipelined frame with statistics: ray counts (identical for every frame of a fixed seed) and a frame's own latency
        barrier()
        c0 = time.perf_counter()
        img, st = step()
        barrier()
        frame_latency_ms = (time.perf_counter() - c0) * 1e3
        rays_local, untraced_local = st["rays"], st.get("rays_untraced", 0)
        kernel_launches.append(st["kernel_launches"])
        logic_ms.append(st["logic_ms"])
        pipe = FramePipeline([render, render2], eye, inv_view, fov, args.width, args.height, rank, world, device)
        for _ in range(max(args.warmup, 2)):   # both handles warm (pool and radiance buffers allocated)
            pipe.submit()
        pipe.drain()
        pipe.done.clear(); pipe.kernel_ms.clear()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.submit()
        pipe.drain()
        barrier()
        elapsed = time.perf_counter() - t0
        assert len(pipe.done) == args.steps
        img = pipe.done[-1]
        kernel_ms = list(pipe.kernel_ms)
        kernel_ms_missing = pipe.kernel_ms_missing
    else:
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            img, st = step()
            kernel_ms.append(st["kernel_ms"])
            logic_ms.append(st["logic_ms"])
            kernel_launches.append(st["kernel_launches"])
            rays_local = st["rays"]
            untraced_local = st.get("rays_untraced", 0)
        barrier()
        elapsed = time.perf_counter() - t0
    # device memory of the per-path radiance of the timed frames on this rank (crt_radiance_storage): one 16-byte value per path of a
    # chunk by default; CRT_FLAG_BOUNDED_RADIANCE would make it a ring of samples (docs/experiments.md section 9)
    radiance_storage = None
    if not multi:
        try:
            rb, rr = render.radiance_storage()
            radiance_storage = {"bytes": rb, "ring_samples": rr}
        except Exception:
            radiance_storage = None
    red_dev = torch.device("cpu") if one_device else device
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    r = torch.tensor([float(rays_local), float(untraced_local)], dtype=torch.float64, device=red_dev)
    # per-rank view of a step (for reading a scaling curve): this rank's mean kernel time per frame, gathered from all ranks
    k_local = float(np.mean([k for k in kernel_ms if k is not None])) if any(k is not None for k in kernel_ms) else 0.0
    k_all = torch.zeros(max(1, world), dtype=torch.float64, device=red_dev)
    k_all[rank if world > 1 else 0] = k_local
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(k_all, op=dist.ReduceOp.SUM)
    rank_kernel_ms = [float(v) for v in k_all.tolist()]
    if multi and mr is not None and mr.rank_stats:
        rank_kernel_ms = [float(s_["kernel_ms"]) for s_ in mr.rank_stats]
    elapsed = float(t.item())
    rays_frame, untraced_frame = float(r[0].item()), float(r[1].item())
    if multi:   # (one process: its counters are the frame's)
        untraced_frame = float(untraced_local)
    ms_per_step = elapsed * 1e3 / args.steps
    # Two rates.  `value` counts the rays that WERE TRACED: next-event samples whose contribution is exactly zero are
    # answered without traversal (same frame bit for bit) and are a ray of the reference but no work of this kernel.  The rate over the
    # reference's own ray count ("one call of DeviceBVH::intersect") is `value_reference_rays`.
    mrays_ref = rays_frame * args.steps / elapsed / 1e6
    mrays = (rays_frame - untraced_frame) * args.steps / elapsed / 1e6
    n_gpus = args.gpus
    single = n_gpus == 1

    if rank == 0:
        wl = args.workload_id          # "c2" .. "c5" when the run is exactly that BASELINE configuration, else None
        c2 = wl == "c2"
        # ---- contract figure: bytes per ray from the counting kernels on a spp=8 slice of the same frame; the same two renders
        #      give the FAST == REFERENCE cross-check (every pixel of the slice, float bits) ----
        def visit_bytes(st, node_bytes=64.0):
            return (node_bytes * st["inner_pops"] + 8.0 * st["leaf_pops"] + 36.0 * st["tri_tests"] + 16.0 * st["hits"]) / st["rays"]

        render.set_spp(8)
        render.traversal = crt.TRAVERSAL_REFERENCE
        render.run_view(eye, inv_view, fov, stats=True, want_mean=True, width=args.width, height=args.height)
        ref_mean, ref_rgb, ref_rays = render.mean_buffer.copy(), render.frame_buffer.copy(), render.stats["rays"]
        b_ray = visit_bytes(render.stats)
        ref_visits = {k: round(render.stats[k] / render.stats["rays"], 2) for k in ("inner_pops", "leaf_pops", "tri_tests", "hits")}
        # the two production modes on the same slice: their visit sets, and their frames against REFERENCE's (every pixel, float bits)
        vs_reference, visits = {}, {}
        b_ray_visited = None
        for mname, mode in (("exact", crt.TRAVERSAL_EXACT), ("fast", crt.TRAVERSAL_FAST)):
            render.traversal = mode
            render.run_view(eye, inv_view, fov, stats=True, want_mean=True, width=args.width, height=args.height)
            visits[mname] = {k: round(render.stats[k] / render.stats["rays"], 2) for k in ("inner_pops", "leaf_pops", "tri_tests", "hits")}
            if mode == trav or b_ray_visited is None:
                b_ray_visited = visit_bytes(render.stats, 112.0)  # the 4-wide tree: 4 boxes + 4 refs per node
            vs_reference[mname] = {
                "sample": "%s %dx%d spp=8, all pixels" % (args.scene, args.width, args.height),
                "pixels_differ_f32_bits": int(np.count_nonzero(np.any(render.mean_buffer.view(np.uint32) != ref_mean.view(np.uint32), axis=2))),
                "rgb8_mismatch": int(np.count_nonzero(np.any(render.frame_buffer != ref_rgb, axis=2))),
                "rays_equal": bool(render.stats["rays"] == ref_rays)}
        fast_vs_reference, exact_vs_reference = vs_reference["fast"], vs_reference["exact"]
        render.traversal = trav
        launches = max(1, int(np.mean(kernel_launches)))
        k_ms_total = float(np.mean(kernel_ms))           # sum of the kernel's launch durations of one frame (rank 0 / slowest rank)
        k_ms = k_ms_total / launches                     # average launch duration
        rays_launch = float(rays_local) / launches
        contract = rays_launch * b_ray / (k_ms * 1e-3) / 1e9
        contract_visited = rays_launch * b_ray_visited / (k_ms * 1e-3) / 1e9
        roofline = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                    "kernel": "k_mega3", "launches_per_frame": launches, "avg_launch_ms": round(k_ms, 4),
                    "kernel_ms_per_frame": round(k_ms_total, 3), "rays_per_launch": int(rays_launch),
                    "compulsory_bytes_per_launch": int(16 * args.width * args.height * args.spp / max(1, launches) / n_gpus),
                    "contract_bytes_per_ray": round(b_ray, 1), "contract_achieved": round(contract, 2),
                    "contract_frac": round(contract / HBM_PEAK_GBPS, 4), "contract_unit": "GB/s",
                    "contract_note": "SURVEY 8(d): algorithmic bytes of the REFERENCE traversal's visit set / kernel time; the kernel "
                                     "does not perform that work (4-wide SAH tree over the same leaves, any-hit visibility rays, untraced zero-contribution samples), so this is not a bound",
                    "reference_visits_per_ray": ref_visits, "exact_visits_per_ray": visits["exact"], "fast_visits_per_ray": visits["fast"],
                    "visited_bytes_per_ray": round(b_ray_visited, 1), "visited_achieved": round(contract_visited, 2)}
        pmc, pmc_note = (load_pmc(wl) if (wl in ("c2", "c3") and single) else (None, "PMC passes exist for the C2 / C3 workloads on one GPU only"))
        if pmc is not None:
            bounds = bounds_from_pmc(pmc, k_ms * 1e-3, load_census(wl))
            roofline["bounds"] = bounds
            if bounds:
                name, f = pick_bound(bounds)
                if name == "valu_issue" and "valu_busy_hw" in bounds[name]:
                    # the vector pipes' busy cycles against all SIMD cycles (the hardware's counter): achieved / peak in SIMD-cycles per second
                    roofline.update({"bound": name, "achieved": round(f * N_SIMDS * CLOCK_GHZ, 1), "peak": round(N_SIMDS * CLOCK_GHZ, 1),
                                     "unit": "G SIMD-cycles/s busy (SQ_ACTIVE_INST_VALU x 4)", "frac": f})
                elif name == "vmem_ta":
                    # the texture addressers' busy cycles against all their cycles (256 of them, one per CU): achieved / peak in TA-cycles per second
                    roofline.update({"bound": name, "achieved": round(f * 256 * CLOCK_GHZ, 1), "peak": round(256 * CLOCK_GHZ, 1),
                                     "unit": "G TA-cycles/s busy (TA_TA_BUSY: the vector-memory pipe)", "frac": f})
                elif name is not None:
                    roofline.update({"bound": name, "achieved": bounds[name]["achieved"], "peak": bounds[name]["peak"],
                                     "unit": bounds[name]["unit"], "frac": bounds[name]["frac"]})
            if "memory_side" in bounds:
                roofline["traffic"] = bounds["memory_side"]["bytes_per_launch"]
                roofline["traffic_note"] = "L2 <-> fabric bytes (Infinity-Cache hits included; no DRAM-only counter is exposed): bounds.memory_side"
            try:
                ev = json.load(open(os.path.join(ROOT, "profiles", "r06_experiments", "sensitivity.json")))
                roofline["sensitivity"] = {"source": "profiles/r06_experiments/sensitivity.json (A/B pairs of round 6, each on one box)", "reading": ev["reading"],
                                           "rows": [{"change": r_["change"], "effect": r_["effect"]} for r_ in ev["rows"]],
                                           "phase_share_of_wave_cycles": {k_: v_ for k_, v_ in ev["phase_share_of_wave_cycles"].items() if k_ != "veach_mis_spp256"}}
            except Exception:
                pass
            roofline["pmc"] = {"src_hash": B.source_hash(), "collected": pmc.get("collected"), "profiled_launch_ms": pmc.get("avg_launch_ms"),
                               "salu_per_valu": pmc.get("salu_per_valu"), "wait_any_frac": pmc.get("SQ_WAIT_ANY/WAVE_CYCLES"),
                               "tcc_miss_frac": pmc.get("tcc_miss_frac")}
            roofline["note"] = ("counts from rocprofv3 --pmc passes of this exact workload and code (hash-stamped), time from HIP events of this "
                                "run; every fraction is against a datasheet peak (MI355X_MICROARCH.md); the unit that is busiest by the hardware's own counters names the bound (pick_bound).  The "
                                "kernel is co-limited: its vector pipes are `bounds.valu_issue.valu_busy_hw` busy (the hardware's counter) "
                                "executing an opcode mix that costs `mix_cycles_per_instr` cycles per instruction instead of the guide's 2, "
                                "`arith_share` of those instructions are box / triangle arithmetic; its waves wait on memory `pmc.wait_any_frac` "
                                "of their cycles (DESIGN.md section 5)")
        else:
            roofline["note"] = pmc_note

        # ---- the same frame with every next-event sample traced (CRT_FLAG_TRACE_ALL): the default path answers the samples whose
        #      contribution is exactly zero without traversal (same frame bit for bit; they still count as rays of the reference) ----
        all_traced = None
        if single and not multi:
            render.set_spp(args.spp)
            render.extra_flags = crt.FLAG_TRACE_ALL
            step()
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            _, st_all = step()
            torch.cuda.synchronize(device)
            dt_all = time.perf_counter() - c0
            render.extra_flags = 0
            all_traced = {"ms_per_step": round(dt_all * 1e3, 3), "mrays_per_sec": round(st_all["rays"] / dt_all / 1e6, 2),
                          "kernel_ms": round(st_all["kernel_ms"], 3)}

        # ---- the same frame in the OTHER of the two production modes, timed and compared with the frame of the timed steps:
        #      CRT_TRAVERSAL_EXACT (default; provably the reference's frame; docs/experiments.md 4.3) against CRT_TRAVERSAL_FAST (+ distance pruning) ----
        other_mode = None
        other_name = None
        if single and not multi and trav in (crt.TRAVERSAL_FAST, crt.TRAVERSAL_EXACT):
            other = crt.TRAVERSAL_EXACT if trav == crt.TRAVERSAL_FAST else crt.TRAVERSAL_FAST
            other_name = "exact_mode" if other == crt.TRAVERSAL_EXACT else "fast_mode"
            render.set_spp(args.spp)
            render.traversal = trav
            own_img, _ = step()
            own_img = own_img.clone() if hasattr(own_img, "clone") else np.array(own_img)
            render.traversal = other
            step()
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            ot_img, st_ot = step()
            torch.cuda.synchronize(device)
            dt_ot = time.perf_counter() - c0
            render.traversal = trav
            same = bool((ot_img == own_img).all()) if hasattr(ot_img, "all") else None
            other_mode = {"traversal": "exact" if other == crt.TRAVERSAL_EXACT else "fast", "ms_per_step": round(dt_ot * 1e3, 3),
                          "kernel_ms": round(st_ot["kernel_ms"], 3), "mrays_per_sec": round(st_ot["rays"] / dt_ot / 1e6, 2),
                          "rgb8_frame_equals_timed_frame": same, "rays_equal": bool(st_ot["rays"] == rays_local)}

        # ---- C3 (veach-mis 800x600 spp=1024: divergence stress) timed by the same run ----
        c3 = None
        if single and c2 and not multi and not args.no_c3:
            t3 = crt.Task(os.path.join(ROOT, "scenes", "veach-mis", "config.json"), base_dir=ROOT)
            s3 = crt.Scene.from_task(t3, 800, 600)
            r3 = crt.Render(s3, 1024, t3.P_RR, t3.light_sample_n, device=local_rank)
            r3.seed = args.seed
            r3.traversal = trav
            iv3 = crt.get_inverse_view_matrix(t3.eye_pos, t3.lookat, t3.up)
            f3 = crt.fov_to_radians(t3.fov_y)

            def step3():
                return render_sharded(r3, t3.eye_pos, iv3, f3, 800, 600, 0, 1, device)

            step3()
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            k3 = []
            for _ in range(3):
                _, st3 = step3()
                k3.append(st3["kernel_ms"])
            torch.cuda.synchronize(device)
            dt3 = (time.perf_counter() - c0) / 3
            r3.extra_flags = crt.FLAG_TRACE_ALL
            step3()
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            _, st3a = step3()
            torch.cuda.synchronize(device)
            dt3a = time.perf_counter() - c0
            r3.extra_flags = 0
            other3 = crt.TRAVERSAL_FAST if trav == crt.TRAVERSAL_EXACT else crt.TRAVERSAL_EXACT
            r3.traversal = other3
            step3()
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            step3()
            torch.cuda.synchronize(device)
            dt3e = time.perf_counter() - c0
            r3.traversal = trav
            c3 = {"workload": "veach-mis 800x600 spp=1024 P_RR=%g light_sample_n=%d" % (float(t3.P_RR), t3.light_sample_n), "frames": 3,
                  "ms_per_frame": round(dt3 * 1e3, 3), "kernel_ms": round(float(np.mean(k3)), 3), "rays_per_frame": int(st3["rays"]),
                  "mrays_per_sec": round((st3["rays"] - st3["rays_untraced"]) / dt3 / 1e6, 2),   # rays traced (as `value`)
                  "mrays_reference_rays_per_sec": round(st3["rays"] / dt3 / 1e6, 2),
                  "untraced_frac": round(st3["rays_untraced"] / st3["rays"], 4),
                  "all_rays_traced_ms": round(dt3a * 1e3, 3), "value_all_rays_traced": round(st3a["rays"] / dt3a / 1e6, 2),
                  ("fast_mode_ms" if other3 == crt.TRAVERSAL_FAST else "exact_mode_ms"): round(dt3e * 1e3, 3)}
            p3, _ = load_pmc("c3")
            if p3 is not None:
                c3["bounds"] = bounds_from_pmc(p3, float(np.mean(k3)) * 1e-3, load_census("c3"))
            r3.free()

        # ---- mesh size: rounds 1-3 rendered the stand-in with 16-bit stack entries that hold leaf refs too (the coupled
        #      pool: <= 32 768 leaf records) and anything larger 9 % slower.  The default is now the decoupled-leaves pool, whose layout
        #      does not depend on the number of leaves (four-wide nodes < 32 768: about 160 000 triangles).  Timed here: the stand-in
        #      under the default, under the coupled pool with 16-bit entries (the round-3 default) and with 32-bit entries (what a larger
        #      mesh got in rounds 1-3), and a 102 412-triangle variant of the scene (scenes/gen_cornell_box.py --detail 6,5), same
        #      camera, same spp ----
        large = None
        if single and c2 and not multi and not args.no_large_scene:
            def timed_frames(rr, n=3):
                render_sharded(rr, eye, inv_view, fov, args.width, args.height, 0, 1, device)
                torch.cuda.synchronize(device)
                c0_ = time.perf_counter()
                ks = []
                for _ in range(n):
                    _, st_ = render_sharded(rr, eye, inv_view, fov, args.width, args.height, 0, 1, device)
                    ks.append(st_["kernel_ms"])
                torch.cuda.synchronize(device)
                return (time.perf_counter() - c0_) * 1e3 / n, float(np.mean(ks)), st_

            def with_env(env, fn):
                old = {k: os.environ.get(k) for k in ("CRT_DEC", "CRT_REF16", "CRT_REF32")}
                for k in old:
                    os.environ.pop(k, None)
                os.environ.update(env)
                try:
                    return fn()
                finally:
                    for k, v in old.items():
                        os.environ.pop(k, None)
                        if v is not None:
                            os.environ[k] = v
            render.set_spp(args.spp)
            render.traversal = trav
            ms_coupled16, k_coupled16, _ = with_env({"CRT_DEC": "0"}, lambda: timed_frames(render))
            ms_coupled32, k_coupled32, _ = with_env({"CRT_DEC": "0", "CRT_REF16": "0"}, lambda: timed_frames(render))
            ms_default, k_default, _ = with_env({}, lambda: timed_frames(render))
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "scenes"))
            import gen_cornell_box
            with tempfile.TemporaryDirectory() as td:
                obj_l, mtl_l, n_tri_l = gen_cornell_box.write_variant(td, (6, 5))
                sl = crt.Scene(args.width, args.height)
                sl.add_obj(obj_l, mtl_l)
                sl.set_BVH(task.bvh_thresh_n)
                rl = crt.Render(sl, args.spp, task.P_RR, task.light_sample_n, device=local_rank)
                rl.seed = args.seed
                rl.traversal = trav
                ai_l = rl.accel_info()
                ms_l, k_l, st_l = with_env({}, lambda: timed_frames(rl))
                ms_l32, k_l32, _ = with_env({"CRT_DEC": "0"}, lambda: timed_frames(rl))
                rl.free()
            large = {"standin_default_layout_ms": round(ms_default, 3), "standin_coupled_16bit_layout_ms": round(ms_coupled16, 3),
                     "standin_coupled_32bit_layout_ms": round(ms_coupled32, 3),
                     "layouts": "default: decoupled leaves, six 16-bit stack levels of inner nodes in LDS + a leaf queue (four-wide nodes < 32 768, "
                                "any number of leaves -- the stand-in and a mesh of the reference's size render with the same kernel); coupled 16-bit "
                                "(CRT_DEC=0; the default of rounds 2-3): eight 16-bit levels, node and leaf refs < 32 768; coupled 32-bit "
                                "(CRT_DEC=0 CRT_REF16=0): four LDS levels (what a larger mesh rendered with in rounds 1-3)",
                     "large_scene": {"workload": "cornell-box --detail 6,5: %d triangles, %d leaf records, %d four-wide nodes, %dx%d spp=%d"
                                                 % (n_tri_l, ai_l["n_leaves"], ai_l["n_nodes4"], args.width, args.height, args.spp),
                                     "layout_caps": ai_l["layout_caps"], "ms_per_frame": round(ms_l, 3), "kernel_ms": round(k_l, 3),
                                     "rays_per_frame": int(st_l["rays"]), "mrays_per_sec": round(st_l["rays"] / ms_l / 1e3, 2),
                                     "coupled_32bit_layout_ms": round(ms_l32, 3)}}

        # ---- CPU baseline + same-run parity gate: the oracle renders the 800x600 spp=--cpu-spp frame on one host
        #      thread; the GPU renders the same frame; every pixel is compared ----
        cpu = None
        parity = None
        if single and not args.no_cpu_baseline:
            import oracle_lib as O
            osc = O.OracleScene(task.OBJ_paths, task.bvh_thresh_n)
            c0 = time.perf_counter()
            orgb, omean, _, ost = osc.render(eye, inv_view, fov, args.width, args.height, args.cpu_spp, task.P_RR,
                                             task.light_sample_n, seed=args.seed)
            cdt = time.perf_counter() - c0
            cpu = {"value": round(ost["rays"] / cdt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                   "sample": "%s %dx%d spp=%d = %d x BASELINE config C1 (the same frame at spp=2: a superset of it) (%d paths, %d rays) in %.1f s, single thread"
                             % (args.scene, args.width, args.height, args.cpu_spp, args.cpu_spp // 2, ost["paths"], ost["rays"], cdt),
                   "ms_per_frame_extrapolated": round(cdt * 1e3 * args.spp / args.cpu_spp, 1)}
            render.set_spp(args.cpu_spp)
            render.traversal = trav
            render.run_view(eye, inv_view, fov, stats=False, want_mean=True, width=args.width, height=args.height)
            gm, gr = render.mean_buffer, render.frame_buffer
            diff = np.abs(gm.astype(np.float64) - omean.astype(np.float64))
            both_nan = np.isnan(gm) & np.isnan(omean)
            diff[both_nan] = 0.0
            diff[np.isnan(diff)] = np.inf
            parity = {"sample": "%s %dx%d spp=%d, all %d pixels, GPU (%s traversal) vs single-thread CPU oracle"
                                % (args.scene, args.width, args.height, args.cpu_spp, args.width * args.height, args.traversal),
                      "tolerance": 1e-5,
                      "pixels_gt_1e-5": int(np.count_nonzero(np.any(diff > 1e-5, axis=2))),
                      "max_abs": float(diff.max()),
                      "pixels_differ_f32_bits": int(np.count_nonzero(np.any((gm.view(np.uint32) != omean.view(np.uint32)) & ~both_nan, axis=2))),
                      "rgb8_mismatch": int(np.count_nonzero(np.any(gr != orgb, axis=2))),
                      "rays_equal": bool(render.stats["rays"] == ost["rays"]),
                      "rays": int(ost["rays"])}
        if args.save_png and img is not None:
            from PIL import Image
            Image.fromarray(img.cpu().numpy()).save(args.save_png)
        rccl_ranks = 0
        rccl_evidence = None
        if multi:
            rccl_ranks = int(mr.info["rccl_ranks"])   # ncclCommCount of the communicator crt_multi_render gathers on (0: peer copies)
            rccl_evidence = {"source": "ncclCommCount (crt_multi_info.rccl_ranks)", "fallback_reason": mr.info.get("fallback_reason") or None}
        elif world > 1 and backend == "nccl":
            rccl_ranks = int(collective_proof["ranks_counted_by_all_reduce"])
            rccl_evidence = dict(collective_proof, source="sum of a device tensor of ones all-reduced over the RCCL process group (torch exposes no ncclCommCount); "
                                                         "distinct_devices = physical-device keys (host name + UUID / PCI address) all-gathered over the same group")
        line = {
            "metric": "Mrays/sec", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %dx%d spp=%d P_RR=%g light_sample_n=%d (stand-in cornell-box.obj, 40972 triangles)"
                                   % (args.scene, args.width, args.height, args.spp, float(task.P_RR), task.light_sample_n)
                       if args.scene == "cornell-box" else
                       "%s %dx%d spp=%d P_RR=%g light_sample_n=%d" % (args.scene, args.width, args.height, args.spp,
                                                                      float(task.P_RR), task.light_sample_n),
                       "traversal": args.traversal, "parallelism": "pixel-tiles x%d" % n_gpus, "seed": args.seed,
                       "engine": "one process, crt_multi_render" if multi else "one process per GPU, torch.distributed",
                       "untraced_samples": "next-event samples whose contribution is exactly zero are answered without traversal "
                                           "(frame bit-identical, still counted as rays of the reference): %.1f%% of the rays%s; "
                                           "`all_rays_traced` times the same frame with every one of them traced"
                                           % (100.0 * untraced_frame / max(1.0, rays_frame), "")},
            "value_definition": "rays TRACED per second (reference rays minus the zero-contribution next-event samples answered without traversal); "
                                "`value_reference_rays` divides the reference's own ray count by the same time; ms_per_step is the figure that compares across implementations",
            "value_reference_rays": round(mrays_ref, 2),
            "rays_traced_per_frame": int(rays_frame - untraced_frame),
            "frames_per_sec": round(1e3 / ms_per_step, 4),
            "frames_in_flight": 2 if pipelined else 1,
            "frames_without_kernel_time": kernel_ms_missing,
            "frame_latency_ms": round(frame_latency_ms, 3) if frame_latency_ms is not None else round(ms_per_step, 3),
            "rays_per_frame": int(rays_frame),
            "rays_definition": "calls of the reference's closest-hit query DeviceBVH::intersect (SURVEY 8(d)); counted by the kernel, "
                               "equal to the oracle's count",
            "collective": {"backend": ("rccl" if rccl_ranks else ("copy" if multi and n_gpus > 1 else (backend or "none"))),
                           "rccl_ranks": rccl_ranks, "rccl_evidence": rccl_evidence, "proof": collective_proof,
                           "launched_by": "torch.distributed.run" if (world > 1 and not os.environ.get("CRT_BENCH_SPAWNED")) else
                                          ("bench.py (self-started ranks)" if world > 1 else "single process")},
            "build_flags": B.built_flags() if not os.environ.get("CRT_LIB_PATH") else "unknown (CRT_LIB_PATH)",
            "library": os.environ.get("CRT_LIB_PATH") or "in-tree cudaraytracing_amd/lib/libcrt.so",
            "src_hash": B.source_hash() if not os.environ.get("CRT_LIB_PATH") else None,
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("CRT_")},
            "workload_id": wl,
            "radiance_storage": radiance_storage,
            "scene_setup": setup,
            "rays_untraced_per_frame": int(untraced_frame),
            "all_rays_traced": all_traced,
            # the same numbers with EVERY ray of the reference traced (no zero-contribution samples answered without traversal):
            # the figures to compare with tracers that count only rays they trace
            "value_all_rays_traced": all_traced["mrays_per_sec"] if all_traced else None,
            "ms_all_rays_traced": all_traced["ms_per_step"] if all_traced else None,
            **({other_name: other_mode} if other_name else {}),
            "mpaths_per_sec": round(args.width * args.height * args.spp * args.steps / elapsed / 1e6, 2),
            "roofline": roofline,
            "exact_vs_reference": exact_vs_reference,
            "fast_vs_reference": fast_vs_reference,
        }
        if n_gpus > 1:
            # what a rank's step is made of: its kernel (min / max over the ranks: the interleaved tiles balance the load, the spread
            # says how well), and what is not kernel time -- the all-gather, the de-interleave, the work-item order pass, the accumulate
            # kernel and launch gaps.  predicted_share_ms: the one-GPU model of DESIGN.md 8 (every launch ends with its waves running
            # their pools dry: a fixed tail, the rest divides by N).
            line["per_rank"] = {"kernel_ms_min": round(min(rank_kernel_ms), 3), "kernel_ms_max": round(max(rank_kernel_ms), 3),
                                "kernel_ms_mean": round(float(np.mean(rank_kernel_ms)), 3), "kernel_ms": [round(v, 3) for v in rank_kernel_ms],
                                "non_kernel_ms_per_step": round(ms_per_step - max(rank_kernel_ms), 3)}
            try:
                sm = json.load(open(os.path.join(ROOT, "profiles", "share_model.json"))).get("workloads", {}).get(wl)
                if sm:
                    line["per_rank"]["predicted_share_ms"] = round(sm["tail_ms"] + (sm["one_gpu_kernel_ms"] - sm["tail_ms"]) / n_gpus, 3)
                    line["per_rank"]["predicted_share_model"] = sm.get("note")
                    line["per_rank"]["measured_share_on_one_gpu_ms"] = sm.get("measured_share_kernel_ms", {}).get(str(n_gpus))
            except Exception:
                pass
        if multi:
            line["multi_info"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in mr.info.items()}
        if parity is not None:
            line["parity"] = parity
        if c3 is not None:
            line["c3"] = c3
        if large is not None:
            line["mesh_size"] = large
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    render.free()
    if render2 is not None:
        render2.free()
    if mr is not None:
        mr.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be positive")
    if args.engine == "procs" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    main_rank(args)


if __name__ == "__main__":
    main()
