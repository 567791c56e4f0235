"""Brick decomposition with fixed-capacity arrays: px x py (x 1) ranks, forward ghost-position halo, NOTHING read back.

The successor of ``domain.SlabDomain`` (which stays as the variable-length, host-planned twin the tests compare with):

* **Fixed capacity, inert rows.**  Local rows live in two segments of fixed size -- ``[0, cap_int)``: particles with no ghost
  neighbor, ``[cap_int, cap)``: particles within r_ghost of a face, ordered by class -- each padded with INERT rows (x = NaN: never
  a neighbor, in no cell, empty neighbor row, zero velocity); the ghosts of every neighbor message have a fixed region behind them.
  ``System.N`` / ``n_ghost`` are the CAPACITIES and never change: no launch is sized by a count the host would have to read, no
  array is re-allocated, every address a step touches is fixed -- a rebuild has no ``.cpu()`` and a whole check period of the
  decomposed step, rebuild included, can be captured into one hipGraph.  Particle counts, message counts and class boundaries
  live in a device vector (``counts``, include/htf_standin.h HTFS_BC_*); overflow of a capacity raises one rebuild late (flags
  copied to pinned memory behind the rebuild), as the device-decided single-domain list reports a neighbor-row overflow.
* **Bricks.**  Up to two decomposed axes: every rank talks to its 2 (slabs) or 8 (bricks: faces + edges) neighbors directly, one
  grouped exchange per halo; a 4 x 2 cut of the 131 072-particle box keeps interior rows where 8 slabs have none (6.72 < 2 x 3.4).
* **Replica mode.**  One rank that is its own neighbor in every direction -- a p-fold periodic replication of its brick, messages
  shifted by a brick width -- so the decomposed step of an 8-rank geometry (rows, ghosts, messages, launches) runs, and is
  timed, on the one GPU a box of this pool has.

Local order inside the boundary segment: class key k_0 + 4 k_1 with k_d = 0 away from both faces of axis d, 1 near the low face
only, 2 near both, 3 near the high face only -- for slabs [interior | left only | both | right only], SlabDomain's order, and
inside a class [stayed | arrived from the highest offset ... lowest], SlabDomain's [stayed | from right | from left]: a run under
BrickDomain(grid = (p, 1, 1)) is bit-identical to the same run under SlabDomain (tests/test_gpu_brick.py).

What HOOMD's Communicator does for the reference under MPI (migrateParticles, exchangeGhosts, the per-step ghost update;
test_mpi_tensorflow.py:57-79); outside the drop-in boundary.
"""
import ctypes as C
import itertools
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

TAG_BASE = 40


def _offsets(ndim):
    """Neighbor offsets in message order: index = sum_d (o_d + 1) 3^d, the centre skipped."""
    out = []
    for raw in range(3 ** ndim):
        o = tuple((raw // 3 ** d) % 3 - 1 for d in range(ndim))
        if any(o):
            out.append(o)
    return out


class BrickDomain:
    fixed_capacity = True

    def __init__(self, system, rank, grid, r_ghost, r_buff=0.4, fractions=None, group=None, transport="torch", replica=False,
                 coords=None, n_global=None, margin=1.25, backend="auto", local_grid=None, replan_every=1):
        """``grid``: (px, py, pz) ranks per box axis, pz = 1 (at most two decomposed axes).  ``fractions``: per axis, the interior
        cut fractions (``comm.decomposition(x=[0.33])`` style) or None for even cuts.  ``replica``: this one rank is every brick
        of ``grid`` (see the module docstring); ``coords``: its brick coordinate then (default: the middle one).
        ``transport``: "torch" (torch.distributed grouped P2P: RCCL on the GPU box, gloo in the CPU tests), "native"
        (libhtf_amd.so's own RCCL communicator: capturable into a hipGraph) or "local" (replica mode: the pack kernel writes the
        ghosts itself).  ``backend``: "kernels" (csrc/brick.hip), "torch" (the restatement, any device) or "auto".
        ``local_grid`` (default: on for device arrays): ghosts are delivered NEXT TO the brick -- a halo message that crosses the
        periodic boundary is shifted by the box vector, as HOOMD wraps its ghosts -- so the neighbor list can be binned on a
        cell grid of the brick + ghost layer alone (``nlist_box()``), not of the whole box: at 8 ranks 7/8 of the global grid's
        cells are empty on any one rank, and under weak scaling the global grid grows with the rank count.  Off: ghosts keep
        their owner's coordinates (SlabDomain's convention; the neighbor ORDER then equals SlabDomain's, bit for bit).
        ``replan_every`` = k > 1: only every k-th rebuild() migrates and re-plans; the others leave rows, classes and messages as
        they are and only refresh the halo -- the list above them is rebuilt on positions that have drifted for up to k periods
        since the plan, so the ghost layer is (k - 1) r_buff thicker than ``r_ghost`` (a pair within the list radius at any of
        those rebuilds had its ghost within r_ghost + (k - 1) r_buff of the face when the plan was made: each side moves at most
        r_buff / 2 per period) and the local cell grid (k - 1) r_buff wider on each side."""
        s = self.sys = system
        self.rank = int(rank)
        self.replica = bool(replica)
        self.group = group
        grid = tuple(int(g) for g in grid) + (1,) * (3 - len(grid))
        if len(grid) != 3 or min(grid) < 1:
            raise ValueError("grid must be (px, py, pz) with positive entries")
        self.grid = grid
        self.axes = [d for d in range(3) if grid[d] > 1]
        if not 1 <= len(self.axes) <= 2:
            raise ValueError("BrickDomain decomposes one or two axes (got grid %r)" % (grid,))
        self.ndim = len(self.axes)
        self.world = 1 if self.replica else int(np.prod(grid))
        if not self.replica and not 0 <= self.rank < self.world:
            raise ValueError("rank %d outside the %r grid" % (self.rank, grid))
        self.replan_every = int(replan_every)
        if self.replan_every < 1:
            raise ValueError("replan_every must be >= 1")
        self.r_buff = float(r_buff)
        self.r_plan = float(r_ghost)                                          # what the caller asked for: the list radius
        self.r_ghost = float(r_ghost) + (self.replan_every - 1) * self.r_buff   # what the plan uses
        self._rebuild_index = 0      # rebuild() calls so far; call i re-plans when i % replan_every == 0
        self._next_light = None      # BrickRun, while capturing: force the kind of the next rebuild (True / False)
        self.n_light = 0
        lo, hi = np.asarray(s.box3x3[0], dtype=np.float64), np.asarray(s.box3x3[1], dtype=np.float64)
        self.L = hi - lo
        if self.replica:
            coords = tuple(coords) if coords is not None else tuple(g // 2 for g in grid)
        else:
            coords = (self.rank % grid[0], (self.rank // grid[0]) % grid[1], self.rank // (grid[0] * grid[1]))
        self.coords = coords
        self.bounds = []
        for d in range(3):
            f = None if fractions is None else (fractions.get(d) if isinstance(fractions, dict) else fractions[d])
            if f is None:
                cuts = np.linspace(0.0, 1.0, grid[d] + 1)
            else:
                cuts = np.concatenate([[0.0], np.asarray(f, dtype=np.float64), [1.0]])
                if len(cuts) != grid[d] + 1 or np.any(np.diff(cuts) <= 0):
                    raise ValueError("axis %d needs %d increasing cut fractions in (0, 1)" % (d, grid[d] - 1))
            self.bounds.append(lo[d] + cuts * (hi[d] - lo[d]))
        for d in self.axes:
            thinnest = float(np.min(np.diff(self.bounds[d])))
            if grid[d] == 2 and not self.replica and thinnest < 2.0 * self.r_ghost:
                raise ValueError("two bricks along axis %d thinner than 2 * r_ghost: both faces lead to the same peer, a particle "
                                 "near both would arrive there twice" % d)
            if thinnest < self.r_ghost:
                raise ValueError("brick thinner than r_ghost along axis %d: ghosts would have to come from beyond the adjacent brick" % d)
        self.lo = np.array([self.bounds[d][coords[d]] for d in range(3)])
        self.hi = np.array([self.bounds[d][coords[d] + 1] for d in range(3)])
        if self.replan_every > 1:
            for d in self.axes:   # (the nearest image must be THE image: brick + ghost layer + drift margin within one period)
                span = (self.hi[d] - self.lo[d]) + 2.0 * (self.r_ghost + (self.replan_every - 1) * self.r_buff)
                if span > self.L[d]:
                    raise ValueError("replan_every=%d needs brick + ghost layer + margin (%.2f) within the box length (%.2f) along axis %d"
                                     % (self.replan_every, span, self.L[d], d))
        self.xlo, self.xhi = float(self.lo[0]), float(self.hi[0])
        self.offsets = _offsets(self.ndim)
        self.n_msg = len(self.offsets)
        self.neighbors = [self._neighbor_rank(o) for o in self.offsets]
        # ---- capacities
        if n_global is None:
            n_global = s.N * (int(np.prod(grid)) if self.replica else 1)
            if not self.replica:
                t = torch.tensor([s.N], dtype=torch.int64)
                if dist.get_backend(group) == "nccl":
                    t = t.to(s.device)
                dist.all_reduce(t, group=group)
                n_global = int(t.item())
        self.n_global = int(n_global)
        rho = self.n_global / float(np.prod(self.L))
        wmax = np.array([float(np.max(np.diff(self.bounds[d]))) for d in range(3)])   # the largest brick: capacities are global
        vol = float(np.prod(wmax))
        inner = float(np.prod([max(wmax[d] - 2.0 * self.r_ghost, 0.0) if d in self.axes else wmax[d] for d in range(3)]))

        def room(n):  # expected count -> capacity: margin, six standard deviations, a floor, whole 64-row groups
            return int(math.ceil((n * margin + 6.0 * math.sqrt(max(n, 1.0)) + 64.0) / 64.0)) * 64

        self.cap_int = room(rho * inner)
        self.cap_bnd = room(rho * (vol - inner))
        self.cap = self.cap_int + self.cap_bnd
        if s.N > self.cap:
            raise ValueError("%d particles do not fit this brick's capacity of %d rows" % (s.N, self.cap))
        ghost_cap, mig_cap = [], []
        for o in self.offsets:
            ext = [wmax[d] for d in range(3)]
            face = [wmax[d] for d in range(3)]
            for k, d in enumerate(self.axes):
                if o[k] != 0:
                    ext[d] = min(self.r_ghost, wmax[d])
                    face[d] = min(self.r_buff * self.replan_every, wmax[d])   # what can cross a face between two PLANS (< r_buff / 2 each way and period)
            ghost_cap.append(room(rho * float(np.prod(ext))))
            mig_cap.append(1 + room(rho * float(np.prod(face))))
        self.ghost_cap = ghost_cap
        self.ghost_off = [int(v) for v in np.concatenate([[0], np.cumsum(ghost_cap)[:-1]])]
        self.mig_cap = mig_cap
        self.mig_off = [int(v) for v in np.concatenate([[0], np.cumsum(mig_cap)[:-1]])]
        self.n_ghost_cap = int(sum(ghost_cap))
        self.mig_rows = int(sum(mig_cap))
        self.cand = self.cap + self.mig_rows
        # ---- shifts.  Replica mode: a message to offset o appears at its receiver moved by -o_d * width_d (halo and migration;
        # a migrant is then wrapped back into the logical box).  Local grid, real ranks: a HALO message that crosses the periodic
        # boundary is moved by the box vector so that its ghosts sit next to the receiving brick.
        self.local_grid = bool(s.pos.is_cuda if local_grid is None else local_grid)
        self.shift = np.zeros((self.n_msg, 3))
        self.mig_shift = np.zeros((self.n_msg, 3))
        for m, o in enumerate(self.offsets):
            for k, d in enumerate(self.axes):
                if self.replica:
                    self.shift[m, d] = self.mig_shift[m, d] = -o[k] * (self.hi[d] - self.lo[d])
                elif self.local_grid and not 0 <= coords[d] + o[k] < grid[d]:
                    self.shift[m, d] = -o[k] * self.L[d]
        self.halo_wrap = self.replica and not self.local_grid
        self.mig_wrap = self.replica
        # ---- adopt the system: fixed-capacity arrays, the particles it came with in the first rows
        dev, dt = s.pos.device, s.pos.dtype
        self.kernels = backend == "kernels" or (backend == "auto" and s.pos.is_cuda and os.environ.get("HTF_DOMAIN_TORCH") != "1")
        if self.kernels and not s.pos.is_cuda:
            raise ValueError("backend='kernels' needs device tensors")
        n0 = s.N
        pos = torch.full((self.cap + self.n_ghost_cap, 4), float("nan"), dtype=dt, device=dev)
        pos[:, 3] = 0
        pos[:n0] = s.pos[:n0]
        vel = torch.zeros((self.cap, 4), dtype=dt, device=dev)
        vel[:, 3] = 1.0
        vel[:n0] = s.vel[:n0]
        s.pos, s.vel = pos, vel
        s.force = torch.zeros((self.cap, 4), dtype=dt, device=dev)
        s.virial = torch.zeros(6 * self.cap, dtype=dt, device=dev)
        s.N, s.n_ghost = self.cap, self.n_ghost_cap
        self.counts = torch.zeros(_lib.BC_WORDS, dtype=torch.int32, device=dev)
        self.mig_send = torch.zeros((self.mig_rows, 8), dtype=dt, device=dev)
        self.mig_recv = torch.zeros((self.mig_rows, 8), dtype=dt, device=dev)
        self.halo_send = torch.empty((self.n_ghost_cap, 4), dtype=dt, device=dev)
        self._n_neigh = None
        self.after_replan = []       # callables run (enqueued) behind every full rebuild (standin.FusedStep: the twin array's ghost tails)
        self.n_interior = self.cap_int            # rows [0, n_interior) have no ghost neighbor (Context.compute_forces_overlapped)
        self.n_rebuilds = 0
        self._works = None
        self._flags_host = None
        self._flags_event = None
        self._native = None
        self.transport = transport
        if transport == "local" and not self.replica:
            raise ValueError("transport='local' delivers a rank's messages to itself: replica mode only")
        if transport not in ("torch", "native", "local", "peer"):
            raise ValueError("transport must be 'torch', 'native', 'local' or 'peer'")
        if self.kernels:
            self._make_device_state()
        if transport == "native":
            self._make_native()
        if transport == "peer":
            self._make_peer()

    # ------------------------------------------------------------------ geometry helpers
    def _neighbor_rank(self, o):
        if self.replica:
            return self.rank
        c = list(self.coords)
        for k, d in enumerate(self.axes):
            c[d] = (c[d] + o[k]) % self.grid[d]
        return c[0] + self.grid[0] * (c[1] + self.grid[1] * c[2])

    def _msg_index(self, o):
        return self.offsets.index(tuple(o))

    def _opposite(self, m):
        return self.n_msg - 1 - m

    @property
    def pending(self):
        return self._works is not None

    @property
    def n_classes(self):
        return 4 ** self.ndim

    def _shifted(self, P, shift, wrap):
        """Positions as a message carries them: shifted and, if asked, wrapped back into the global box (brick.hip ``shifted``)."""
        if not np.any(shift):
            return P
        P = P.clone()
        for c in range(3):
            if shift[c] != 0.0:
                dt, dev = P.dtype, P.device
                x = P[:, c] + torch.as_tensor(float(shift[c]), dtype=dt, device=dev)
                if wrap:
                    lo = torch.as_tensor(float(self.sys.box3x3[0][c]), dtype=dt, device=dev)
                    L = torch.as_tensor(float(self.L[c]), dtype=dt, device=dev)
                    x = x - torch.floor((x - lo) * (1.0 / L)) * L
                P[:, c] = x
        return P

    def nlist_box(self):
        """The box the neighbor list is binned and searched on: with a local grid the brick + its ghost layer, NOT periodic
        along the decomposed axes (the ghosts are real images next to the brick); else the global box."""
        if not self.local_grid:
            return self.sys.box3x3, self.sys.periodic
        b = np.array(self.sys.box3x3, dtype=np.float64)
        per = list(self.sys.periodic)
        for d in self.axes:
            pad = (self.replan_every - 1) * self.r_buff   # (rows and ghosts drift between plans: the grid must still hold them)
            b[0][d], b[1][d] = self.lo[d] - self.r_ghost - pad, self.hi[d] + self.r_ghost + pad
            per[d] = 0
        return b, tuple(per)

    def image_lengths(self):
        """The logical box length along every decomposed axis -- the period of the coordinates there -- when the list may be binned on
        rows that have drifted since the plan (a local grid and replan_every > 1: the binning then takes every coordinate as its image
        nearest the grid); zeros otherwise (right after a re-plan every row and ghost already sits next to the brick)."""
        on = self.local_grid and self.replan_every > 1
        return tuple(float(self.L[d]) if (on and d in self.axes) else 0.0 for d in range(3))

    def _msg_takes_class(self, m, c):
        for k in range(self.ndim):
            o = self.offsets[m][k]
            kd = (c >> (2 * k)) & 3
            if o == -1 and kd not in (1, 2):
                return False
            if o == 1 and kd not in (2, 3):
                return False
        return True

    # ------------------------------------------------------------------ device state
    def _make_device_state(self):
        s = self.sys
        dev, dt = s.pos.device, s.pos.dtype
        g = self.geom = _lib.Brick()
        g.ndim, g.n_msg, g.replica = self.ndim, self.n_msg, int(self.replica)
        for k, d in enumerate(self.axes):
            g.axis[k], g.p[k], g.me[k] = d, self.grid[d], self.coords[d]
        g.r_ghost = self.r_ghost
        g.cap_int, g.cap_bnd = self.cap_int, self.cap_bnd
        for m in range(self.n_msg):
            g.ghost_cap[m], g.ghost_off[m] = self.ghost_cap[m], self.ghost_off[m]
            g.mig_cap[m], g.mig_off[m] = self.mig_cap[m], self.mig_off[m]
            for c in range(3):
                g.shift[m][c] = float(self.shift[m, c])
                g.mig_shift[m][c] = float(self.mig_shift[m, c])
        g.halo_wrap, g.mig_wrap = int(self.halo_wrap), int(self.mig_wrap)
        for c in range(3):
            g.box_lo[c], g.box_L[c] = float(self.sys.box3x3[0][c]), float(self.L[c])
        b = np.zeros((2, _lib.BRICK_MAX_P + 1))
        for k, d in enumerate(self.axes):
            if self.grid[d] > _lib.BRICK_MAX_P:
                raise ValueError("at most %d bricks along an axis" % _lib.BRICK_MAX_P)
            b[k, :self.grid[d] + 1] = self.bounds[d]
        self._bounds_dev = torch.as_tensor(b, dtype=dt, device=dev).contiguous()
        tiles = (self.cand + 1023) // 1024
        self._wk = {"key": torch.empty(self.cand, dtype=torch.int32, device=dev),
                    "order": torch.empty(self.cand, dtype=torch.int32, device=dev),
                    "scratch": torch.zeros(32 * tiles, dtype=torch.int32, device=dev),
                    "start1": torch.zeros(17, dtype=torch.int32, device=dev),
                    "start2": torch.zeros(33, dtype=torch.int32, device=dev),
                    "tmp_pos": torch.empty((self.cand, 4), dtype=dt, device=dev),
                    "tmp_vel": torch.empty((self.cand, 4), dtype=dt, device=dev)}
        w = self.work = _lib.BrickWork()
        w.key, w.order, w.sort_scratch = self._wk["key"].data_ptr(), self._wk["order"].data_ptr(), self._wk["scratch"].data_ptr()
        w.start1, w.start2 = self._wk["start1"].data_ptr(), self._wk["start2"].data_ptr()
        w.tmp_pos, w.tmp_vel = self._wk["tmp_pos"].data_ptr(), self._wk["tmp_vel"].data_ptr()
        self._flags_host = torch.zeros(_lib.BC_WORDS, dtype=torch.int32).pin_memory()

    def _make_native(self):
        from .domain import _NativeHalo
        if self.replica:
            self._native = _NativeHalo.shared(0, 1, None, solo=True)
        else:
            self._native = _NativeHalo.shared(self.rank, self.world, self.group)

    def _make_peer(self):
        """Transport "peer": NO communication library anywhere in a run.  Every rank owns one block of device memory its
        neighbors' kernels store into -- fine-grained (hipExtMallocWithFlags(hipDeviceMallocFinegrained): coherent between agents
        while kernels run, what a mapping across xGMI wants; HTF_PEER_MEMORY=coarse: plain hipMalloc) -- holding
        * the halo inbox ([2][ghost rows] Scalar4: two halves, by the exchange number's parity) and its signal words ({sequence,
          rows} per incoming message): the packing kernels of its NEIGHBORS store into them, htfs_brick_unpack_halo copies what has
          arrived into the ghost region (csrc/brick.hip *_peer_kernel);
        * the migration mailbox ([2][migration rows] of 8 scalars) and its signal words: a re-plan's messages (csrc/mailbox.hip
          htfs_mailbox_push / _pull -- until round 6 they travelled through torch.distributed / RCCL);
        * the all-reduce table ([world][2] 8-byte words): the distance check's maximum over the ranks in one launch
          (htfs_mailbox_allreduce_max_f32).
        The block travels as a hipIpc handle through the process group at construction (the same memory for a replica rank); state
        words stay local.  Everything is ordinary kernel work on ONE stream: whole check periods, re-plan included, capture into
        hipGraphs (standin.BrickRun) with nothing but this transport."""
        s = self.sys
        if not self.kernels:
            raise ValueError("transport='peer' needs the kernels backend")
        if getattr(self, "peer", None) is not None:
            return
        dev, dt = s.pos.device, s.pos.dtype
        esz = 4 * (8 if dt == torch.float64 else 4)                   # bytes of a Scalar4
        world = self.world

        def up(n, a=256):
            return (int(n) + a - 1) // a * a
        lay, off = {}, 0
        for name, nbytes in (("inbox", 2 * self.n_ghost_cap * esz), ("signal", 2 * _lib.BRICK_MAX_MSG * 4),
                             ("mig", 2 * self.mig_rows * 2 * esz), ("mig_signal", _lib.MBOX_MAX_MSG * 4),
                             ("reduce", max(world, 1) * 2 * 8)):
            lay[name] = off
            off = up(off + nbytes)
        self._peer_layout, total = lay, off
        want_fine = os.environ.get("HTF_PEER_MEMORY", "fine") != "coarse"
        ptr = C.c_void_p()
        rc = _lib.lib.htfs_shared_alloc(total, 1 if want_fine else 0, C.byref(ptr))
        self.peer_memory = "fine-grained (hipExtMallocWithFlags)" if want_fine else "coarse-grained (hipMalloc)"
        if rc != 0 and want_fine:
            why = _lib.last_error()
            _lib.check(_lib.lib.htfs_shared_alloc(total, 0, C.byref(ptr)))
            self.peer_memory = "coarse-grained (hipMalloc; the fine-grained allocation failed: %s)" % why
        elif rc != 0:
            _lib.check(rc)
        self._peer_block = ptr.value
        self._peer_state = torch.zeros(12, dtype=torch.int32, device=dev)   # halo [4] | migration [4] | all-reduce [4]
        torch.cuda.synchronize(dev)
        blocks = {self.rank: self._peer_block}
        self._peer_opened = []
        if not self.replica and world > 1:
            h = (C.c_char * _lib.IPC_HANDLE_BYTES)()
            _lib.check(_lib.lib.htfs_ipc_export(self._peer_block, h))
            everybody = [None] * world
            dist.all_gather_object(everybody, (bytes(h.raw), os.getpid()), group=self.group)
            for q in range(world):                       # (the all-reduce table is all-to-all: every rank maps every block)
                if q == self.rank:
                    continue
                raw, pid = everybody[q]
                if pid == os.getpid():
                    raise RuntimeError("two ranks in one process cannot map each other's memory through IPC")
                p = C.c_void_p()
                _lib.check(_lib.lib.htfs_ipc_import((C.c_char * _lib.IPC_HANDLE_BYTES).from_buffer_copy(raw), C.byref(p)))
                blocks[q] = p.value
                self._peer_opened.append(p.value)
        self._peer_blocks = blocks
        spin = int(os.environ.get("HTF_PEER_SPIN", str(1 << 22)))   # x ~1 us a poll: seconds (ranks start skewed), then a FLAG -- never a hang
        st = self._peer_state.data_ptr()
        pr = self.peer = _lib.Peer()
        mb = self.mailbox = _lib.Mailbox()
        for m in range(self.n_msg):
            base = blocks[self.neighbors[m]]
            pr.inbox[m], pr.signal[m] = base + lay["inbox"], base + lay["signal"]
            mb.remote[m] = base + lay["mig"]
            mb.remote_signal[m] = base + lay["mig_signal"] + 4 * self._opposite(m)   # my message m is its message from offset -m
        pr.my_inbox, pr.my_signal, pr.state = self._peer_block + lay["inbox"], self._peer_block + lay["signal"], st
        pr.spin_limit = spin
        mb.mine, mb.my_signal, mb.state = self._peer_block + lay["mig"], self._peer_block + lay["mig_signal"], st + 16
        mb.spin_limit = spin
        unit_rows = 2 * esz // 16                                    # 16-byte units per migration row (position + velocity)
        mb.half_units = self.mig_rows * unit_rows
        U = C.c_uint * _lib.MBOX_MAX_MSG
        # message m: rows [mig_off[m], + mig_cap[m]) of mig_send -> the region of source offset -m in the receiver's mailbox, which is
        # where the receiver's pull finds "the message from the neighbor at offset index j = opposite(m)": rows [mig_off[j], ...)
        self._mb_send_off = U(*[self.mig_off[m] * unit_rows for m in range(self.n_msg)])
        self._mb_units = U(*[self.mig_cap[m] * unit_rows for m in range(self.n_msg)])
        self._mb_box_off = U(*[self.mig_off[self._opposite(m)] * unit_rows for m in range(self.n_msg)])
        self._mb_recv_off = U(*[self.mig_off[j] * unit_rows for j in range(self.n_msg)])
        self._mb_recv_box = U(*[self.mig_off[j] * unit_rows for j in range(self.n_msg)])
        self._mb_row_units = unit_rows
        for m in range(self.n_msg):
            if self.mig_cap[m] != self.mig_cap[self._opposite(m)]:
                raise RuntimeError("migration capacities are not symmetric")   # (they are global: the largest brick's)
        rb = self.reduce_box = _lib.ReduceBox()
        ranks = [self.rank] if self.replica else list(range(world))
        for k, q in enumerate(ranks):
            rb.remote[k] = blocks[q] + lay["reduce"]
        rb.mine, rb.state, rb.spin_limit = self._peer_block + lay["reduce"], st + 32, spin
        rb.world, rb.rank = len(ranks), (0 if self.replica else self.rank)
        if not self.replica and world > 1:
            dist.barrier(group=self.group)               # nobody stores into a block that is not mapped everywhere yet

    def __del__(self):
        try:
            for p in getattr(self, "_peer_opened", []):
                _lib.lib.htfs_ipc_close(p)
            if getattr(self, "_peer_block", None):
                _lib.lib.htfs_shared_free(self._peer_block)
                self._peer_block = None
        except Exception:  # noqa: BLE001  (interpreter teardown)
            pass

    def allreduce_max(self, value):
        """value[0] <- its maximum over the ranks, on the current stream: the library's RCCL communicator (transport "native") or one
        launch through the peers' tables (transport "peer"); None when this domain has no device-side all-reduce of its own."""
        if self._native is not None and self.transport != "peer":
            self._native.allreduce_max(value)
            return True
        if getattr(self, "reduce_box", None) is not None and self.transport == "peer":
            _lib.check(_lib.lib.htfs_mailbox_allreduce_max_f32(C.byref(self.reduce_box), value.data_ptr(), self.counts.data_ptr() + 4 * _lib.BC_FLAGS,
                                                               _lib.BF_HALO_TIMEOUT, self._stream()))
            return True
        if self._native is not None:
            self._native.allreduce_max(value)
            return True
        return None

    def enable_row_slots(self):
        """The table the one-kernel step's epilogue reads (standin.FusedStep): the slot of every boundary row in every halo message,
        refreshed by every re-plan from now on."""
        if getattr(self, "row_slots", None) is None:
            self.row_slots = torch.full((max(self.cap_bnd, 1), _lib.BRICK_MAX_MSG), -1, dtype=torch.int32, device=self.sys.pos.device)
            self._fill_row_slots()

    def _fill_row_slots(self):
        _lib.check(_lib.lib.htfs_brick_row_slots(C.byref(self.geom), self.counts.data_ptr(), self.row_slots.data_ptr(), self._stream()))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.sys.pos.device).cuda_stream)

    def attach_n_neigh(self, n_neigh):
        """The neighbor list's row counts: a rebuild zeroes the rows that became inert (the search only writes rows of particles)."""
        self._n_neigh = n_neigh

    # ------------------------------------------------------------------ flags / counts (reports; they synchronise)
    def _raise_flags(self, f):
        if not f:
            return
        what = [t for bit, t in ((_lib.BF_LOST, "a particle crossed more than one brick between neighbor-list rebuilds"),
                                 (_lib.BF_MIG_OVERFLOW, "more migrants than a migration message holds"),
                                 (_lib.BF_INT_OVERFLOW, "more interior particles than cap_int = %d" % self.cap_int),
                                 (_lib.BF_BND_OVERFLOW, "more boundary particles than cap_bnd = %d" % self.cap_bnd),
                                 (_lib.BF_GHOST_OVERFLOW, "more ghosts than a halo message holds"),
                                 (_lib.BF_HALO_TIMEOUT, "a neighbor's halo message did not arrive (transport 'peer': are all ranks stepping?)"))
                if f & bit]
        raise RuntimeError("BrickDomain (rank %d): %s; construct it with a larger margin" % (self.rank, "; ".join(what)))

    def poll_flags(self):
        """The overflow / lost-particle flags of the PREVIOUS rebuild (copied to pinned memory behind it: no wait)."""
        if self._flags_event is not None:
            self._flags_event.synchronize()
            self._flags_event = None
            self._raise_flags(int(self._flags_host[_lib.BC_FLAGS]))

    def counts_host(self):
        """The device counts, read now (synchronises: reports and tests only)."""
        c = self.counts.cpu().numpy().astype(np.int64)
        self._raise_flags(int(c[_lib.BC_FLAGS]))
        return c

    @property
    def n_local(self):
        c = self.counts_host()
        return int(c[_lib.BC_N_INT] + c[_lib.BC_N_BND])

    @property
    def n_migrated(self):
        return int(self.counts_host()[_lib.BC_N_ARRIVED])

    @property
    def n_ghosts(self):
        c = self.counts_host()
        return int(c[_lib.BC_MSG:_lib.BC_MSG + self.n_msg].sum())   # (symmetric bricks: what I send is what I receive)

    @property
    def local_counts(self):
        """Particles per rank (an all-gather and a read-back: the training step's global batch under batching only)."""
        n = torch.tensor([self.n_local], dtype=torch.int64)
        if self.world == 1:
            return [int(n)]
        if dist.get_backend(self.group) == "nccl":
            n = n.to(self.sys.pos.device)
        out = [torch.empty_like(n) for _ in range(self.world)]
        dist.all_gather(out, n, group=self.group)
        return [int(v.item()) for v in out]

    def live_rows(self):
        """Indices of the rows that hold a particle (synchronises)."""
        return torch.nonzero(~torch.isnan(self.sys.pos[:self.cap, 0])).flatten()

    def row_classes(self):
        raise NotImplementedError("the particle sorter is not supported under BrickDomain")

    # ------------------------------------------------------------------ transports
    def _exchange(self, send, recv, caps, offs, tag0, overlap=False):
        """Message m = rows [offs[m], offs[m] + caps[m]) of ``send`` to the neighbor at offset m; the message FROM the neighbor at
        offset index j lands in rows [offs[j], +caps[j]) of ``recv``.  Sends are posted in ascending offset order, receives in
        descending order: between any two ranks the k-th send meets the k-th receive (RCCL matches by order, not by tag)."""
        if self.replica and self._native is None:   # (a replica rank without RCCL: "local" and "peer")
            pairs = [(m, self._opposite(m)) for m in range(self.n_msg)]
            if send.is_cuda and send.shape[0] == recv.shape[0] and all(caps[m] == caps[j] for m, j in pairs):
                # one gather instead of n_msg slice copies (eight dependent nodes of a captured 2-D rebuild)
                key = (tuple(int(c) for c in caps), tuple(int(o) for o in offs))
                perm = self._local_perm.get(key) if hasattr(self, "_local_perm") else None
                if perm is None:
                    rows = torch.arange(recv.shape[0], dtype=torch.int64)
                    for m, j in pairs:
                        rows[offs[j]:offs[j] + caps[j]] = torch.arange(offs[m], offs[m] + caps[m], dtype=torch.int64)
                    if not hasattr(self, "_local_perm"):
                        self._local_perm = {}
                    perm = self._local_perm[key] = rows.to(send.device)
                torch.index_select(send, 0, perm, out=recv)
                return []
            for m, j in pairs:
                recv[offs[j]:offs[j] + caps[j]] = send[offs[m]:offs[m] + caps[m]]
            return []
        if self._native is not None:
            self._native.exchange(send, recv, caps, offs, self.neighbors, self._opposite, overlap=overlap)
            return [self._native]
        ops = []
        for m in range(self.n_msg):
            ops.append(dist.P2POp(dist.isend, send[offs[m]:offs[m] + caps[m]], self.neighbors[m], self.group, tag0 + m))
        for j in reversed(range(self.n_msg)):
            # what my neighbor at offset j sent me is ITS message to offset -j: tagged with that index
            ops.append(dist.P2POp(dist.irecv, recv[offs[j]:offs[j] + caps[j]], self.neighbors[j], self.group, tag0 + self._opposite(j)))
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            from .domain import _StagedHalo
            return [_StagedHalo(ops)]
        return dist.batch_isend_irecv(ops)

    # ------------------------------------------------------------------ rebuild
    def rebuild(self):
        """Communicator::migrateParticles + exchangeGhosts: call before every neighbor-list build."""
        s = self.sys
        capturing = getattr(self, "_capturing", False)   # inside a hipGraph capture (standin.BrickRun): no host-side waits
        light = self._next_light if self._next_light is not None else (self._rebuild_index % self.replan_every != 0)
        self._rebuild_index += 1
        if light:
            # rows, classes and messages as the last plan left them; only this step's halo (the list is rebuilt by the caller)
            self.n_rebuilds += 1
            self.n_light += 1
            self.exchange()
            return
        if self.kernels:
            if not capturing:
                self.poll_flags()
            g, w = C.byref(self.geom), C.byref(self.work)
            _lib.check(_lib.lib.htfs_brick_migrate_pack(g, s.pos.data_ptr(), s.vel.data_ptr(), s.scalar_code, self._bounds_dev.data_ptr(),
                                                        w, self.mig_send.data_ptr(), self.counts.data_ptr(), self._stream()))
            if self.transport == "peer":
                # the migration messages through the neighbors' mailboxes (csrc/mailbox.hip): two launches, no library call
                mb = C.byref(self.mailbox)
                _lib.check(_lib.lib.htfs_mailbox_push(mb, self.n_msg, self.mig_send.data_ptr(), self._mb_send_off, self._mb_units,
                                                      self._mb_box_off, self._mb_row_units, self._stream()))
                _lib.check(_lib.lib.htfs_mailbox_pull(mb, self.n_msg, self.mig_recv.data_ptr(), self._mb_recv_off, self._mb_units,
                                                      self._mb_recv_box, self._mb_row_units, self.counts.data_ptr() + 4 * _lib.BC_FLAGS,
                                                      _lib.BF_HALO_TIMEOUT, self._stream()))
            else:
                for wk in self._exchange(self.mig_send, self.mig_recv, self.mig_cap, self.mig_off, TAG_BASE):
                    wk.wait()
            _lib.check(_lib.lib.htfs_brick_migrate_merge(g, s.pos.data_ptr(), s.vel.data_ptr(), s.scalar_code, self._bounds_dev.data_ptr(),
                                                         w, self.mig_recv.data_ptr(),
                                                         self._n_neigh.data_ptr() if self._n_neigh is not None else None,
                                                         self.counts.data_ptr(), self._stream()))
            if getattr(self, "row_slots", None) is not None:
                self._fill_row_slots()
            for fn in self.after_replan:
                fn()
            if not (capturing and getattr(self, "_mirrored", False)):   # (BrickRun: the next check kernel carries them to the host)
                self._flags_host.copy_(self.counts, non_blocking=True)
            if not capturing:
                self._flags_event = torch.cuda.Event()
                self._flags_event.record(torch.cuda.current_stream(s.pos.device))
        else:
            self._rebuild_torch()
        self.n_rebuilds += 1
        self._packed = False   # (whatever the integrator packed describes the rows of the old plan)
        self.exchange()

    def _dest_keys_torch(self, P):
        """(key per row: 0 stay | 1 + message | n_msg + 1 inert, lost mask) -- brick_dest_kernel's arithmetic."""
        live = ~torch.isnan(P[:, 0])
        raw = torch.zeros(P.shape[0], dtype=torch.int64, device=P.device)
        stay = torch.ones(P.shape[0], dtype=torch.bool, device=P.device)
        lost = torch.zeros_like(stay)
        mul = 1
        for k, d in enumerate(self.axes):
            p, me = self.grid[d], self.coords[d]
            x = P[:, d].contiguous()
            bnd = torch.as_tensor(self.bounds[d], dtype=P.dtype, device=P.device)
            owner = (x[:, None] >= bnd[None, 1:-1]).sum(dim=1)
            if p == 2 and not self.replica:
                off = (owner != me).to(torch.int64)
            elif p == 2:
                Lg = bnd[2] - bnd[0]
                dx = x - 0.5 * (bnd[me] + bnd[me + 1])
                dx = dx - Lg * torch.round(dx / Lg)
                off = torch.where(owner != me, torch.where(dx < 0, -1, 1), 0)
            else:
                left, right = (me - 1) % p, (me + 1) % p
                off = torch.where(owner == me, 0, torch.where(owner == left, -1, torch.where(owner == right, 1, 0)))
                lost |= live & (owner != me) & (owner != left) & (owner != right)
            stay &= off == 0
            raw += (off + 1) * mul
            mul *= 3
        centre = (3 ** self.ndim - 1) // 2
        key = torch.where(stay | lost, torch.zeros_like(raw), 1 + torch.where(raw < centre, raw, raw - 1))
        key = torch.where(live, key, torch.full_like(key, self.n_msg + 1))
        return key, lost

    def _class_keys_torch(self, P):
        k = torch.zeros(P.shape[0], dtype=torch.int64, device=P.device)
        for i, d in enumerate(self.axes):
            x = P[:, d]
            bnd = torch.as_tensor(self.bounds[d], dtype=P.dtype, device=P.device)
            rg = torch.as_tensor(self.r_ghost, dtype=P.dtype, device=P.device)
            near_lo, near_hi = x < bnd[self.coords[d]] + rg, x >= bnd[self.coords[d] + 1] - rg
            kd = torch.where(near_lo, torch.where(near_hi, 2, 1), torch.where(near_hi, 3, 0))
            k += kd << (2 * i)
        return k

    def _rebuild_torch(self):
        """The restatement of csrc/brick.hip in torch ops (CPU tensors in the gloo tests; the kernels are checked against it)."""
        s = self.sys
        cap, dt, dev = self.cap, s.pos.dtype, s.pos.device
        P, V = s.pos[:cap], s.vel[:cap]
        key, lost = self._dest_keys_torch(P)
        flags = 0
        if bool(lost.any()):
            flags |= _lib.BF_LOST
        order = torch.sort(key, stable=True)[1]
        cnt = torch.bincount(key, minlength=self.n_msg + 2).cpu().numpy()
        start = np.concatenate([[0], np.cumsum(cnt)])
        PV = torch.cat([P, V], dim=1)
        words = 1 if dt == torch.float32 else 2
        hdr_send = self.mig_send.view(torch.int32)
        for m in range(self.n_msg):
            n = int(cnt[1 + m])
            room = self.mig_cap[m] - 1
            if n > room:
                flags |= _lib.BF_MIG_OVERFLOW
                n = room
            rows = order[start[1 + m]:start[1 + m] + n]
            rec = PV[rows].clone()
            rec[:, :4] = self._shifted(rec[:, :4], self.mig_shift[m], self.mig_wrap)
            o = self.mig_off[m]
            self.mig_send[o + 1:o + 1 + n] = rec
            hdr_send[o, 0] = n
        del words
        for wk in self._exchange(self.mig_send, self.mig_recv, self.mig_cap, self.mig_off, TAG_BASE):
            wk.wait()
        hdr_recv = self.mig_recv.view(torch.int32)
        parts = [PV[order[:int(cnt[0])]]]
        arrived = 0
        for j in reversed(range(self.n_msg)):
            o = self.mig_off[j]
            n = int(hdr_recv[o, 0])
            arrived += n
            parts.append(self.mig_recv[o + 1:o + 1 + n])
        cand = torch.cat(parts, dim=0)
        k2 = self._class_keys_torch(cand[:, :4])
        order2 = torch.sort(k2, stable=True)[1]
        ccnt = torch.bincount(k2, minlength=self.n_classes).cpu().numpy()
        cstart = np.concatenate([[0], np.cumsum(ccnt)])
        n_int, n_live = int(ccnt[0]), int(cand.shape[0])
        n_bnd = n_live - n_int
        if n_int > self.cap_int:
            flags |= _lib.BF_INT_OVERFLOW
        if n_bnd > self.cap_bnd:
            flags |= _lib.BF_BND_OVERFLOW
        msg = []
        for m in range(self.n_msg):
            n = int(sum(ccnt[c] for c in range(1, self.n_classes) if self._msg_takes_class(m, c)))
            if n > self.ghost_cap[m]:
                flags |= _lib.BF_GHOST_OVERFLOW
            msg.append(min(n, self.ghost_cap[m]))
        self._raise_flags(flags)
        s.pos[:cap] = float("nan")
        s.pos[:cap, 3] = 0
        s.vel[:cap] = 0
        s.vel[:cap, 3] = 1
        srt = cand[order2]
        s.pos[:n_int] = srt[:n_int, :4]
        s.vel[:n_int] = srt[:n_int, 4:]
        s.pos[self.cap_int:self.cap_int + n_bnd] = srt[n_int:, :4]
        s.vel[self.cap_int:self.cap_int + n_bnd] = srt[n_int:, 4:]
        if self._n_neigh is not None:
            self._n_neigh[n_int:self.cap_int] = 0
            self._n_neigh[self.cap_int + n_bnd:] = 0
        c = np.zeros(_lib.BC_WORDS, dtype=np.int64)
        prev = self.counts.cpu().numpy()
        c[_lib.BC_N_INT], c[_lib.BC_N_BND], c[_lib.BC_N_CAND] = n_int, n_bnd, n_live
        c[_lib.BC_N_ARRIVED] = int(prev[_lib.BC_N_ARRIVED]) + arrived
        c[_lib.BC_REBUILDS] = int(prev[_lib.BC_REBUILDS]) + 1
        c[_lib.BC_MSG:_lib.BC_MSG + self.n_msg] = msg
        c[_lib.BC_CLASS:_lib.BC_CLASS + self.n_classes + 1] = cstart
        c[_lib.BC_SLOT:_lib.BC_SLOT + 16 * _lib.BRICK_MAX_MSG] = 0xFFFFFFFF
        for cl in range(1, self.n_classes):
            for m in range(self.n_msg):
                if self._msg_takes_class(m, cl):
                    c[_lib.BC_SLOT + cl * _lib.BRICK_MAX_MSG + m] = sum(int(ccnt[cc]) for cc in range(1, cl) if self._msg_takes_class(m, cc))
        self.counts.copy_(torch.as_tensor((c & 0xFFFFFFFF).astype(np.uint32).view(np.int32)))

    # ------------------------------------------------------------------ per-step halo
    def _pack_halo_torch(self):
        s = self.sys
        c = self.counts.cpu().numpy().astype(np.int64)
        cstart = c[_lib.BC_CLASS:_lib.BC_CLASS + self.n_classes + 1]
        n_int = int(cstart[1])
        self.halo_send[:] = float("nan")
        self.halo_send[:, 3] = 0
        for m in range(self.n_msg):
            rows = [torch.arange(self.cap_int + cstart[cl] - n_int, self.cap_int + cstart[cl + 1] - n_int, device=s.pos.device)
                    for cl in range(1, self.n_classes) if self._msg_takes_class(m, cl)]
            rows = torch.cat(rows)[:self.ghost_cap[m]]
            buf = self._shifted(s.pos[rows], self.shift[m], self.halo_wrap)
            self.halo_send[self.ghost_off[m]:self.ghost_off[m] + len(rows)] = buf

    def nve_step(self, dt):
        """The integrator's step over the local rows AND the halo messages of the new positions in one launch
        (htfs_brick_nve_halo; the kernels backend): the next exchange_begin() finds the messages packed."""
        s = self.sys
        if self.transport == "peer":
            _lib.check(_lib.lib.htfs_brick_nve_halo_peer(C.byref(self.geom), s.pos.data_ptr(), s.vel.data_ptr(), s.force.data_ptr(),
                                                         s.scalar_code, float(dt), C.byref(s.box), self.counts.data_ptr(),
                                                         C.byref(self.peer), self._stream()))
            self._packed = True
            return
        direct = self.transport == "local"
        _lib.check(_lib.lib.htfs_brick_nve_halo(C.byref(self.geom), s.pos.data_ptr(), s.vel.data_ptr(), s.force.data_ptr(), s.scalar_code,
                                                float(dt), C.byref(s.box), self.counts.data_ptr(),
                                                None if direct else self.halo_send.data_ptr(),
                                                s.pos.data_ptr() + self.cap * 4 * s.pos.element_size() if direct else None,
                                                self._stream()))
        self._packed = True

    @property
    def overlaps(self):
        """Does a posted halo travel while the interior rows are evaluated?  Not when the pack kernel delivers it itself, and not
        inside a hipGraph capture (where the RCCL calls stay on the captured stream)."""
        if self.transport == "peer":
            return True     # stores into the neighbors' inboxes: in flight while the interior rows run, inside a capture too
        return self.transport != "local" and not getattr(self, "_capturing", False)

    def exchange_begin(self):
        """Post the per-step forward halo: pack the messages from the boundary segment (unless nve_step() has), one grouped
        exchange into the ghost regions.  Nothing may write ``pos`` until exchange_end()."""
        s = self.sys
        ghosts = s.pos[self.cap:]
        packed, self._packed = getattr(self, "_packed", False), False
        if self.kernels and self.transport == "peer":
            if not packed:
                _lib.check(_lib.lib.htfs_brick_pack_halo_peer(C.byref(self.geom), s.pos.data_ptr(), s.scalar_code, self.counts.data_ptr(),
                                                              C.byref(self.peer), self._stream()))
            self._works = [self]     # exchange_end(): the unpack kernel
            return
        if self.kernels:
            direct = self.transport == "local"
            if not packed:
                _lib.check(_lib.lib.htfs_brick_pack_halo(C.byref(self.geom), s.pos.data_ptr(), s.scalar_code, self.counts.data_ptr(),
                                                         None if direct else self.halo_send.data_ptr(),
                                                         ghosts.data_ptr() if direct else None, self._stream()))
            if direct:
                self._works = []
                return
        else:
            self._pack_halo_torch()
        # (inside a hipGraph capture the RCCL calls stay on the captured stream: on this ROCm, RCCL enqueued on a stream that
        #  JOINED a capture through an event crashes hipStreamEndCapture -- tools/rccl_graph_probe.py, profiles/r05_rccl_graph_probe.txt)
        self._works = self._exchange(self.halo_send, ghosts, self.ghost_cap, self.ghost_off, TAG_BASE + 16,
                                     overlap=not getattr(self, "_capturing", False))

    def wait(self):
        """Transport "peer": the receiving half of an exchange (the stream waits on the device; the host does not)."""
        s = self.sys
        _lib.check(_lib.lib.htfs_brick_unpack_halo(C.byref(self.geom), s.pos.data_ptr(), s.scalar_code, C.byref(self.peer),
                                                   self.counts.data_ptr(), self._stream()))

    def exchange_end(self):
        if self._works is not None:
            for w in self._works:
                w.wait()
            self._works = None

    def exchange(self):
        self.exchange_begin()
        self.exchange_end()
