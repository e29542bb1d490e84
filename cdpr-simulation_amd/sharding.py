"""Multi-GPU placement: robots are independent (the reference simulates exactly one, PLG.cpp:202-246 touches one
model's joints), so a batch shards by contiguous blocks with NO collective on the data path.  One process per GPU;
the rendezvous bench.py's contract asks for (barrier, max of elapsed time, a few gathered words) runs over a plain
socket between the rank processes plus a shared-memory spin barrier for the timed edges: no PyTorch, no RCCL (north_star).
`CDPR_BENCH_BACKEND=nccl|gloo` opts into a torch.distributed process group instead (to answer "does RCCL see N ranks").
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Tuple


def shard_range(rank: int, world: int, total: int) -> Tuple[int, int]:
    """Robots [lo, hi) owned by `rank`: contiguous blocks whose sizes differ by at most one."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedEngine:
    """One process, several GPUs: robots [lo, hi) of the batch live on device d (contiguous blocks, BASELINE config 4:
    524 288 robots = 8 x 65 536).  Every call fans out to the per-device engines; launches are asynchronous, so one
    host thread keeps all GPUs busy and nothing crosses between devices.  Same methods as `Engine`; batched arrays
    are split / concatenated along the robot axis."""

    def __init__(self, config, devices):
        from dataclasses import replace

        from .engine import Engine

        self.config = config
        self.devices = list(devices)
        self.B, self.n = int(config.batch), config.n_cables
        self.spans = [shard_range(i, len(self.devices), self.B) for i in range(len(self.devices))]
        if any(hi - lo < 1 for lo, hi in self.spans):
            raise ValueError("fewer robots than devices")
        self.engines = [Engine(replace(config, batch=hi - lo), device=d) for (lo, hi), d in zip(self.spans, self.devices)]

    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _split(self, a, width):
        import numpy as np

        a = np.asarray(a, dtype=np.float32)
        if a.size == width:  # one message broadcast to every robot
            return [a] * len(self.engines)
        a = a.reshape(self.B, width)
        return [a[lo:hi] for lo, hi in self.spans]

    def set_platform_state(self, pose7=None, twist6=None):
        ps = [None] * len(self.engines) if pose7 is None else self._split(pose7, 7)
        ts = [None] * len(self.engines) if twist6 is None else self._split(twist6, 6)
        for e, p, t in zip(self.engines, ps, ts):
            e.set_platform_state(p, t)

    def _command(self, name, axes, mask):
        import numpy as np

        a = np.asarray(axes, dtype=np.float32)
        if a.size not in (self.n, self.n * self.B):
            return 1  # CDPR_IGNORED, as every shard would answer (PLG.cpp:68-73)
        masks = [None] * len(self.engines) if mask is None else [np.asarray(mask, dtype=np.uint8).reshape(self.B)[lo:hi] for lo, hi in self.spans]
        return max(getattr(e, name)(x, m) for e, x, m in zip(self.engines, self._split(a, self.n), masks))

    def set_velocity_command(self, axes, mask=None):
        return self._command("set_velocity_command", axes, mask)

    def set_position_command(self, axes, mask=None):
        return self._command("set_position_command", axes, mask)

    def set_force_command(self, axes, mask=None):
        return self._command("set_force_command", axes, mask)

    def update(self, nsteps=1, steps_per_launch=1):
        for e in self.engines:  # asynchronous: all devices run concurrently
            e.update(nsteps, steps_per_launch)

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def reset(self):
        for e in self.engines:
            e.reset()

    @property
    def step_count(self):
        return self.engines[0].step_count

    def _gather(self, name):
        import numpy as np

        parts = [getattr(e, name)() for e in self.engines]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(len(parts[0])))

    def rollout_velocity(self, commands, ref_position):
        """MPC fan-out over all devices (BASELINE config 5: 4 096 robots x 128 samples x 64 steps, 512 robots per GPU):
        commands[B, H, S, n], ref_position[B, 3] -> cost[B, S].  Each device rolls out its own robots."""
        import numpy as np

        c = np.asarray(commands, dtype=np.float32)
        ref = np.asarray(ref_position, dtype=np.float32).reshape(self.B, 3)
        # queue on every device first (upload + launch return at once), then collect: the GPUs run concurrently
        launched = []
        try:
            for e, (lo, hi) in zip(self.engines, self.spans):
                e.rollout_launch(c[lo:hi], ref[lo:hi])
                launched.append(e)
        except Exception:
            for e in launched:  # a device that failed to launch must not leave the others with a pending rollout
                e.rollout_discard()
            raise
        return np.concatenate([e.rollout_fetch() for e in self.engines])

    def joint_states(self):
        return self._gather("joint_states")

    def platform_state(self):
        return self._gather("platform_state")

    def observables(self):
        return self._gather("observables")

    def raw_state(self):
        return self._gather("raw_state")

    def fk_state(self):
        return self._gather("fk_state")

    def td_state(self):
        return self._gather("td_state")

    def limit_state(self):
        import numpy as np

        return np.concatenate([e.limit_state() for e in self.engines])


class LocalSpinBarrier:
    """Barrier between the rank processes of ONE node through a few bytes of shared memory (/dev/shm): every rank owns
    one 8-byte slot, writes the barrier's epoch into it and spins until every slot has reached that epoch (single writer
    per slot, aligned 8-byte stores: no atomics needed).  A couple of microseconds, against the ~100 us of a socket, RCCL
    or gloo barrier — which matters when the timed region of a short bench run is a few hundred microseconds long and ends
    with a barrier.  Rank 0 creates the file; the others open it after `group_barrier()` (the rendezvous' own barrier)."""

    def __init__(self, rank: int, world: int, key: str, group_barrier):
        import mmap

        self.rank, self.world, self.epoch = rank, world, 0
        self.timeout_s = 300.0
        self.path = f"/dev/shm/cdpr_bench_barrier_{key}"
        self.ok = True  # every rank runs the same sequence of group calls whatever fails locally (no rank may
        self._f = self._mm = self._slots = None  # drop out of a collective); the caller agrees on `ok` across ranks
        if rank == 0:
            try:
                with open(self.path, "wb") as f:
                    f.write(b"\0" * 8 * world)
            except OSError:
                self.ok = False
        group_barrier()
        try:
            self._f = open(self.path, "r+b")
            self._mm = mmap.mmap(self._f.fileno(), 8 * world)
            import numpy as np

            self._slots = np.frombuffer(self._mm, dtype=np.uint64, count=world)
        except (OSError, ValueError):
            self.ok = False
        if os.environ.get("CDPR_TEST_SPIN_FAIL_RANK") == str(rank):  # test hook: this rank's set-up "failed"
            self.ok = False

    def wait(self) -> None:
        self.epoch += 1
        self._slots[self.rank] = self.epoch
        e = self.epoch
        s = self._slots
        spins = 0
        t_start = 0.0
        while int(s.min()) < e:
            spins += 1
            if (spins & 0xFFFF) == 0:  # a rank that died must not leave the others spinning for ever
                import time

                now = time.monotonic()
                if t_start == 0.0:
                    t_start = now
                elif now - t_start > self.timeout_s:
                    raise RuntimeError(f"LocalSpinBarrier: rank {self.rank} waited {self.timeout_s:.0f} s at epoch {e} (slots {s.tolist()})")

    def close(self) -> None:
        self._slots = None
        for h in (self._mm, self._f):
            try:
                if h is not None:
                    h.close()
            except Exception:
                pass
        self._mm = self._f = None
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


class SocketGroup:
    """The default rendezvous: a star of stream sockets, rank 0 in the middle.  ONE primitive - every rank hands in a short
    byte string and gets every rank's string back, indexed by rank - carries the barrier, the max and the gathers of
    bench.py's contract (a few hundred bytes per call, outside every timed region; the timed edges use LocalSpinBarrier).
    Python's standard library only: a multi-GPU run imports neither torch nor RCCL, so nothing else initialises the GPU that
    is being timed and the first run on eight real devices cannot die in a second runtime.

    Address: all ranks on one node (bench.py's contract; LOCAL_WORLD_SIZE == WORLD_SIZE) meet on an abstract-namespace
    Unix socket named after the uid and MASTER_PORT - no port to reserve (under torch.distributed.run the agent's store
    owns MASTER_PORT itself), nothing left behind in the file system, and two jobs on a node cannot share a MASTER_PORT.
    Across nodes: TCP on MASTER_ADDR, port CDPR_RDV_PORT (default MASTER_PORT + 1)."""

    name = "socket"

    def __init__(self, rank: int, world: int, local_world: int, timeout_s: float = 600.0):  # (generous: a fresh box pages the image in during the first imports)
        import socket
        import time

        self.rank, self.world = rank, world
        self.timeout_s = timeout_s
        port = int(os.environ.get("MASTER_PORT", "29500"))
        if local_world == world and os.environ.get("CDPR_RDV_TCP") != "1":
            family, addr = socket.AF_UNIX, f"\0cdpr_rdv_{os.getuid()}_{port}"
            self.address = "unix:@" + addr[1:]
        else:
            family, addr = socket.AF_INET, (os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("CDPR_RDV_PORT", str(port + 1))))
            self.address = f"tcp:{addr[0]}:{addr[1]}"
        self._peers = {}   # rank 0: rank -> connection
        self._conn = None  # other ranks: the connection to rank 0
        self._srv = None
        deadline = time.monotonic() + timeout_s
        if rank == 0:
            srv = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind(("", addr[1]))
                except OSError as exc:
                    raise RuntimeError(f"cdpr rendezvous: rank 0 cannot listen on port {addr[1]} ({exc}); set CDPR_RDV_PORT to a free port") from exc
            else:
                srv.bind(addr)
            srv.listen(world)
            self._srv = srv
            while len(self._peers) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    missing = sorted(set(range(1, world)) - set(self._peers))
                    raise RuntimeError(f"cdpr rendezvous: ranks {missing} did not connect to {self.address} within {timeout_s:.0f} s") from None
                c.settimeout(timeout_s)
                peer = int.from_bytes(self._recv_exact(c, 4), "little")
                if not 0 < peer < world or peer in self._peers:
                    c.close()
                    raise RuntimeError(f"cdpr rendezvous: unexpected hello from rank {peer}")
                if family == socket.AF_INET:
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[peer] = c
        else:
            while True:  # rank 0 may not be listening yet
                c = socket.socket(family, socket.SOCK_STREAM)
                try:
                    c.connect(addr)
                    break
                except OSError:
                    c.close()
                    if time.monotonic() > deadline:
                        raise RuntimeError(f"cdpr rendezvous: rank {rank} could not reach rank 0 at {self.address} within {timeout_s:.0f} s") from None
                    time.sleep(0.02)
            if family == socket.AF_INET:
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout_s)
            c.sendall(rank.to_bytes(4, "little"))
            self._conn = c

    @staticmethod
    def _recv_exact(c, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise RuntimeError("cdpr rendezvous: a peer closed its connection (did a rank die?)")
            buf += part
        return bytes(buf)

    @classmethod
    def _recv_msg(cls, c) -> bytes:
        return cls._recv_exact(c, int.from_bytes(cls._recv_exact(c, 4), "little"))

    @staticmethod
    def _pack(parts) -> bytes:
        return b"".join(len(p).to_bytes(4, "little") + p for p in parts)

    def all_gather(self, payload: bytes) -> list:
        """Every rank's payload, indexed by rank.  Blocks until all ranks called it (so it is also the barrier)."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv_msg(self._peers[r]) for r in range(1, self.world)]
            blob = self._pack(parts)
            for r in range(1, self.world):
                self._peers[r].sendall(len(blob).to_bytes(4, "little") + blob)
            return parts
        self._conn.sendall(len(payload).to_bytes(4, "little") + payload)
        blob = self._recv_msg(self._conn)
        out, at = [], 0
        for _ in range(self.world):
            k = int.from_bytes(blob[at:at + 4], "little")
            out.append(blob[at + 4:at + 4 + k])
            at += 4 + k
        return out

    def device_sync(self) -> None:  # nothing of this group lives on a GPU
        pass

    def close(self) -> None:
        for c in list(self._peers.values()) + [self._conn, self._srv]:
            try:
                if c is not None:
                    c.close()
            except OSError:
                pass
        self._peers, self._conn, self._srv = {}, None, None


class TorchGroup:
    """Opt-in rendezvous over a torch.distributed process group (CDPR_BENCH_BACKEND=nccl|gloo): same one primitive.  With
    "nccl" (= RCCL on ROCm) the group is pre-flighted so that a run cannot die or hang in it:

    1. every rank states over a TCP store (CDPR_STORE_PORT, reserved by bench.py's own launcher; MASTER_PORT + 1 under
       torch.distributed.run) whether it can try RCCL at all (a GPU of its own visible to torch); one "no" and all take gloo;
    2. else every rank initialises the RCCL group (eagerly: device_id = its GPU) and proves it with one ASYNC all-reduce
       waited for with a timeout - a hang becomes a "no", not a watchdog abort - and states the outcome; one failure and
       every rank tears its group down and takes gloo over the same store.
    The store's timeout is longer than the process group's, and a store timeout counts as "no" instead of killing the rank
    (a rank that came up quickly must outwait a slow peer's group timeout).  `fallback` says what happened."""

    PG_TIMEOUT_S = 120.0
    STORE_TIMEOUT_S = 300.0

    def __init__(self, rank: int, local_rank: int, world: int, local_world: int, backend: str):
        import datetime

        import torch
        import torch.distributed as dist

        self.rank, self.world, self._dist, self._torch = rank, world, dist, torch
        self.fallback = None
        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500"))
        store_port = int(os.environ.get("CDPR_STORE_PORT", str(port + 1)))
        try:
            store = dist.TCPStore(addr, store_port, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=self.STORE_TIMEOUT_S), wait_for_workers=False)
        except Exception as exc:  # noqa: BLE001
            raise RuntimeError(f"cdpr rendezvous: no TCP store on {addr}:{store_port} ({type(exc).__name__}: {exc}); set CDPR_STORE_PORT to a free port") from exc
        self._store = store
        pg_timeout = datetime.timedelta(seconds=self.PG_TIMEOUT_S)

        def agree(key: str, ok: bool) -> bool:
            """True when EVERY rank says ok (each rank publishes its word, then reads everyone's); a peer that never
            publishes within the store's timeout counts as 'no'."""
            store.set(f"{key}/{rank}", b"1" if ok else b"0")
            try:
                return all(store.get(f"{key}/{r}") == b"1" for r in range(world))  # (get blocks until the key is there)
            except Exception:  # noqa: BLE001  (store timeout)
                return False

        # (device_count() does not initialise the GPU.  CDPR_RENDEZVOUS_ASSUME_GPUS=1: take the RCCL branch whatever the count,
        #  which is how the CPU test suite drives the fallback)
        can_try = backend == "nccl" and (torch.cuda.device_count() >= local_world or os.environ.get("CDPR_RENDEZVOUS_ASSUME_GPUS") == "1")
        up = False
        if backend == "nccl" and not agree("can_try", can_try):
            self.fallback = "gloo: a rank has no GPU of its own visible to torch (torch.cuda.device_count() < ranks on the node)"
        elif backend == "nccl":
            err = ""
            try:
                torch.cuda.set_device(local_rank)  # "nccl" is RCCL on ROCm
                dist.init_process_group(backend="nccl", store=dist.PrefixStore("rccl", store), rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
                t = torch.ones(1, device="cuda")
                work = dist.all_reduce(t, async_op=True)
                ok = bool(work.wait(timeout=datetime.timedelta(seconds=60)))
                torch.cuda.synchronize()
                ok = ok and float(t.item()) == float(world)
            except Exception as exc:  # noqa: BLE001  (whatever RCCL / the driver throws: the run goes on over gloo)
                ok, err = False, f"{type(exc).__name__}: {exc}"
            if agree("rccl_up", ok):
                up = True
            else:
                self.fallback = "gloo: the RCCL group did not come up on every rank" + (f" (this rank: {err[:200]})" if err else "")
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
        if not up:
            dist.init_process_group(backend="gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world, timeout=pg_timeout)
        self.name = str(dist.get_backend())

    def all_gather(self, payload: bytes) -> list:
        out = [None] * self.world
        self._dist.all_gather_object(out, payload)
        return out

    def device_sync(self) -> None:
        if self.name == "nccl":
            self._torch.cuda.synchronize()

    def close(self) -> None:
        try:
            self._dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


@dataclass
class RankContext:
    rank: int = 0
    local_rank: int = 0
    world: int = 1
    _group: object = None
    _spin: object = None
    fallback: object = None  # why the rendezvous is not on the requested backend (None: it is)

    @classmethod
    def from_env(cls, backend: str = "socket") -> "RankContext":
        """The rendezvous of a multi-rank run from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (what bench.py's own launcher and
        torch.distributed.run both export).  backend "socket" (default): SocketGroup, standard library only.  "nccl" / "gloo":
        TorchGroup (imports torch; RCCL pre-flighted, gloo when it cannot come up)."""
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1 and os.environ.get("CDPR_FORCE_RENDEZVOUS") != "1":  # (the override: a one-rank group,
            return cls(rank, local_rank, world, None)                          # to exercise the group's calls on one GPU)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if backend in ("nccl", "gloo"):
            group = TorchGroup(rank, local_rank, world, local_world, backend)
        elif backend == "socket":
            group = SocketGroup(rank, world, local_world)
        else:
            raise ValueError(f"unknown rendezvous backend {backend!r} (socket, nccl, gloo)")
        ctx = cls(rank, local_rank, world, group)
        ctx.fallback = getattr(group, "fallback", None)
        if local_world == world:  # every rank on this node (bench.py's contract: one node)
            spin = LocalSpinBarrier(rank, world, f"{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}", ctx._group_barrier)
            # all ranks or none: a rank on its own in the spin barrier (or out of it) would hang the others
            if ctx.min_over_ranks(1.0 if spin.ok else 0.0) > 0.5:
                ctx._spin = spin
            else:
                spin.close()
        return ctx

    def _group_barrier(self) -> None:
        self._group.all_gather(b"")

    def gather_strings(self, text: str) -> list:
        """One short string per rank, indexed by rank (device identities for the report)."""
        if self._group is None:
            return [text]
        return [b.decode() for b in self._group.all_gather(text.encode())]

    def backend_name(self) -> str:
        return "none" if self._group is None else self._group.name

    def barrier(self) -> None:
        if self._group is not None:
            self._group.device_sync()
            self._group_barrier()

    def fast_barrier(self) -> None:
        """Barrier with microsecond latency for the edges of a timed region (shared-memory spin on one node); falls back to
        the group's barrier."""
        if self._spin is not None:
            self._spin.wait()
        elif self._group is not None:
            self._group_barrier()

    def gather_over_ranks(self, values) -> list:
        """values (a short list of floats) of every rank, indexed by rank: what makes a straggler visible next to the
        max-over-ranks figure.  One gather of a few doubles, outside every timed region."""
        import struct

        vals = [float(v) for v in values]
        if self._group is None:
            return [vals]
        parts = self._group.all_gather(struct.pack(f"<{len(vals)}d", *vals))
        return [list(struct.unpack(f"<{len(p) // 8}d", p)) for p in parts]

    def max_over_ranks(self, value: float) -> float:
        return max(v[0] for v in self.gather_over_ranks([value]))

    def min_over_ranks(self, value: float) -> float:
        return -self.max_over_ranks(-value)

    def close(self) -> None:
        if self._group is not None:
            self._group_barrier()
            if self._spin is not None:
                self._spin.close()
                self._spin = None
            self._group.close()
            self._group = None
