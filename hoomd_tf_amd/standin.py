"""Stand-in for the slice of HOOMD-blue the force path sits in: particle data in HOOMD
layout, a binned FULL neighbor list, a leapfrog NVE integrator.

HOOMD-blue is not installable in this image, so ``tfcompute.attach`` binds to these
objects instead of ``hoomd.context.current``.  Everything here is OUTSIDE the drop-in
boundary (include/htf_amd.h): with a real HOOMD the plugin shim of INTEGRATION.md
passes HOOMD's own arrays to the same C entry points.  torch supplies device memory
and the sort used for cell binning; the search/integration kernels are HIP
(csrc/standin.hip).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import lib, check
from .ops import raw_stream


def fcc_positions(n_cells, rho):
    """4*n^3 particles on an fcc lattice at number density rho, box centred on 0."""
    a = (4.0 / rho) ** (1.0 / 3.0)
    L = n_cells * a
    ijk = np.stack(np.meshgrid(*[np.arange(n_cells)] * 3, indexing="ij"), -1).reshape(-1, 3)
    base = np.array([[0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]])
    pos = ((ijk[:, None, :] + base[None]) * a).reshape(-1, 3) - L / 2
    return pos, np.array([L, L, L]), a


def sc_positions(n_cells, rho):
    a = (1.0 / rho) ** (1.0 / 3.0)
    L = n_cells * a
    ijk = np.stack(np.meshgrid(*[np.arange(n_cells)] * 3, indexing="ij"), -1).reshape(-1, 3)
    return (ijk + 0.5) * a - L / 2, np.array([L, L, L]), a


class System:
    """ParticleData + BoxDim analogue: HOOMD-layout device arrays.

    pos   Scalar4 [N + n_ghost]  (x, y, z, int type bits in w)
    vel   Scalar4 [N]            (vx, vy, vz, mass)
    force Scalar4 [N]            (fx, fy, fz, energy)  -- ForceCompute::m_force
    virial Scalar [6 * N]        -- ForceCompute::m_virial, pitch N
    """

    def __init__(self, positions, box_L, types=None, dtype=torch.float32, device="cuda", periodic=(1, 1, 1)):
        positions = np.asarray(positions, dtype=np.float64)
        self.N = int(positions.shape[0])
        self.n_ghost = 0
        self.dtype = dtype
        self.device = torch.device(device)
        L = np.asarray(box_L, dtype=np.float64)
        self.box3x3 = np.array([-L / 2, L / 2, [0.0, 0.0, 0.0]])
        self.periodic = tuple(int(p) for p in periodic)
        self.box = _lib.make_box(self.box3x3, self.periodic)
        types = np.zeros(self.N, dtype=np.int32) if types is None else np.asarray(types, dtype=np.int32)
        self.pos = ops.stuff_types(torch.from_numpy(positions).to(self.device),
                                   torch.from_numpy(types).to(self.device), dtype)
        self.vel = torch.zeros((self.N, 4), dtype=dtype, device=self.device)
        self.vel[:, 3] = 1.0
        self.force = torch.zeros((self.N, 4), dtype=dtype, device=self.device)
        self.virial = torch.zeros(6 * self.N, dtype=dtype, device=self.device)
        self.timestep = 0

    @property
    def scalar_code(self):
        return _lib.HTF_F64 if self.dtype == torch.float64 else _lib.HTF_F32

    def randomize_velocities(self, kT, seed):
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        v = torch.randn((self.N, 3), generator=g, dtype=torch.float64) * math.sqrt(kT)
        v -= v.mean(dim=0, keepdim=True)
        self.vel[:, :3] = v.to(self.dtype).to(self.device)

    def append_particles(self, positions, types):
        """snapshot.particles.resize + restore_snapshot: add particles at the end (zero velocity)."""
        positions = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        types = np.asarray(types, dtype=np.int32)
        if self.n_ghost:
            raise ValueError("cannot add particles to a decomposed system")
        add = ops.stuff_types(torch.from_numpy(positions).to(self.device), torch.from_numpy(types).to(self.device), self.dtype)
        M = int(add.shape[0])
        self.pos = torch.cat([self.pos[: self.N], add], dim=0)
        v = torch.zeros((M, 4), dtype=self.dtype, device=self.device)
        v[:, 3] = 1.0
        self.vel = torch.cat([self.vel, v], dim=0)
        self.N += M
        self.force = torch.zeros((self.N, 4), dtype=self.dtype, device=self.device)
        self.virial = torch.zeros(6 * self.N, dtype=self.dtype, device=self.device)

    def positions_numpy(self):
        return self.pos[: self.N, :3].double().cpu().numpy()

    def types_numpy(self):
        w = self.pos[: self.N, 3].contiguous()
        if self.dtype == torch.float32:
            return w.view(torch.int32).cpu().numpy()
        return (w.view(torch.int64) & 0xFFFFFFFF).to(torch.int32).cpu().numpy()


class DeferredRebuildRule:
    """NeighborList::distanceCheck's decision taken ONE CHECK LATE, for a box decomposed over ranks.

    Under decomposition every rank must take the same decision, so a check costs an all-reduce of the largest
    displacement -- and, read back at once, a drained queue on every rank every ``check_period`` steps (at 16 k rows per
    rank that stall is longer than the force kernel).  Here the all-reduced value of check k is read at check k + 1, when it
    has long arrived, and the rebuild is decided from what is known then: with d the largest displacement since the last
    rebuild (d = 0 there) and g its growth over the last period, rebuild NOW if d + 2 g > r_buff / 2 -- the value two
    periods after the newest measurement, i.e. at the next opportunity to act, extrapolated linearly (the maximum over
    particles of a ballistic-then-diffusive displacement grows sub-linearly, so the line errs on the early side).
    A measurement that already exceeds r_buff / 2 when it is read is counted as a DANGEROUS build, as HOOMD counts them:
    the list had been used beyond its guarantee; lower ``check_period``.  Pure host arithmetic on a handful of floats, shared
    by the asynchronous path and by the synchronous reference path of the tests."""

    def __init__(self, half_buffer, lookahead=2.0):
        self.h = float(half_buffer)
        self.lookahead = float(lookahead)
        self.hist = []
        self.dangerous = 0

    def push(self, d):
        self.hist.append(float(d))
        if d > self.h:
            self.dangerous += 1

    def decide(self):
        if not self.hist:
            return False
        d1 = self.hist[-1]
        d0 = self.hist[-2] if len(self.hist) > 1 else 0.0
        return d1 + self.lookahead * max(d1 - d0, 0.0) > self.h

    def reset(self):
        self.hist = []


class CellNlist:
    """hoomd.md.nlist.cell analogue: FULL neighbor list, fixed pitch head list, rebuilt
    when any particle has moved more than r_buff / 2 (NeighborList::distanceCheck)."""

    def __init__(self, system, r_cut, r_buff=0.4, pitch=None, check_period=1, sort_particles=False,
                 device_decision=False, deferred_reference=False):
        self.sys = system
        # Under domain decomposition ``device_decision`` selects DeferredRebuildRule fed without a host synchronisation
        # (device all-reduce -> pinned copy -> read one check later); ``deferred_reference`` feeds the same rule from a
        # blocking read at every check -- the host-decided twin the tests compare trajectories with.
        self.deferred_reference = bool(deferred_reference)
        self._rule = None
        self._dd_prev = None
        self._dd_i = 0
        # device_decision: after the first build the distance check and the rebuild it may trigger are
        # enqueued together, the binning / search kernels gated on the device by the check's result
        # (htfs_set_gate): the step loop never reads the check back.  The buffers keep their addresses, a
        # neighbor-row overflow (more entries than ``pitch``) is reported one check late.  Single-rank,
        # unsorted systems only; otherwise the host decides as NeighborList::distanceCheck does.
        self.device_decision = bool(device_decision)
        self._stat = self._stat_host = self._stat_event = None
        # HOOMD's SFCPackUpdater analogue: renumber the local particles in cell order at every
        # rebuild, so that a particle's neighbors sit in a few contiguous index runs and the
        # position gathers of the force path coalesce.  Off by default (it changes particle
        # indices, as HOOMD's sorter does; the reference's tests disable it where they track ids).
        self.sort_particles = bool(sort_particles)
        self.r_cut = float(r_cut)
        self.r_buff = float(r_buff)
        self.check_period = int(check_period)
        self.pitch = pitch
        self.n_neigh = self.head_list = self.nlist = None
        self._ref = None
        self._disp = torch.zeros(1, dtype=torch.float32, device=system.device)
        self._max = torch.zeros(1, dtype=torch.int32, device=system.device)
        self.n_builds = 0
        self._subscribers = []
        self.domain = None  # SlabDomain when the box is decomposed over ranks
        self.type_split = -1  # >= 0: no pairs across this type id (mapped beads vs all-atom particles)

    def subscribe(self, rcut_fn):
        """NeighborList r_cut subscription (tensorflowcompute.py:116-120): the list is
        built for the largest cutoff any attached compute asks for."""
        self._subscribers.append(rcut_fn)
        self.r_cut = max([self.r_cut] + [float(fn()) for fn in self._subscribers])
        self._ref = None

    @property
    def r_list(self):
        return self.r_cut + self.r_buff

    def _ncell(self):
        """cells per direction and stencil half-width: cells at least r_list / 2 wide (5 per direction
        searched) when the box holds >= 5 of them, else at least r_list wide (3 searched), else one."""
        b3, _ = self._search_box()
        L = b3[1] - b3[0]
        n = np.ones(3, dtype=int)
        w = np.zeros(3, dtype=int)
        for d in range(3):
            fine = int(np.floor(L[d] / (self.r_list / 2.0)))
            coarse = int(np.floor(L[d] / self.r_list))
            if fine >= 5:
                n[d], w[d] = fine, 2
            elif coarse >= 3:
                n[d], w[d] = coarse, 1
        return n, w

    def _image_lengths(self):
        """The period of the coordinates along the axes the domain's local grid is not periodic on (htfs_rebuild_nlist_ghosts'
        image_L), or None."""
        fn = getattr(self.domain, "image_lengths", None)
        if fn is None:
            return None
        key = fn()
        if getattr(self, "_image_key", None) != key:
            self._image_key = key
            self._image_arr = (C.c_double * 3)(*key) if any(key) else None
        return self._image_arr

    def _search_box(self):
        """(box3x3, periodic) the list is binned and searched on: the system's box, or -- under a BrickDomain with a local cell
        grid -- the brick + its ghost layer, not periodic along the decomposed axes (brick.BrickDomain.nlist_box)."""
        if self.domain is not None and getattr(self.domain, "local_grid", False):
            return self.domain.nlist_box()
        return self.sys.box3x3, self.sys.periodic

    def build(self):
        s = self.sys
        if self.domain is not None:
            self.domain.rebuild()  # Communicator: migrate particles, re-plan + fill ghosts
        b3, per = self._search_box()
        if getattr(self, "_sbox_key", None) != (b3.tobytes(), per):
            self._sbox_key = (np.asarray(b3).tobytes(), per)
            self._sbox = _lib.make_box(np.asarray(b3), per)
        sbox = self._sbox
        Ntot = s.N + s.n_ghost
        n, w = self._ncell()
        n3 = (C.c_int * 3)(*[int(x) for x in n])
        w3 = (C.c_int * 3)(*[int(x) for x in w])
        ncell = int(n[0] * n[1] * n[2])
        stream = C.c_void_p(raw_stream(s.device.index))
        if getattr(self, "_scr_n", None) != (Ntot, ncell, int(w[1]), int(w[2])):
            self._scr_n = (Ntot, ncell, int(w[1]), int(w[2]))
            self._cell_of = torch.empty(Ntot, dtype=torch.int32, device=s.device)
            self._order = torch.empty(Ntot, dtype=torch.int32, device=s.device)
            self._cell_start = torch.empty(ncell + 1, dtype=torch.int32, device=s.device)
            self._bin_scratch = torch.empty(2 * ncell, dtype=torch.int32, device=s.device)
            self._pos_sorted = torch.empty((Ntot, 4), dtype=s.pos.dtype, device=s.device)
            # candidate ranges per (cell, stencil row): this list's own table (htf_standin.h HTFS_RANGE_WORDS), part of what a
            # captured step carries by address
            self._ranges = torch.empty(4 * ncell * int((2 * w[1] + 1) * (2 * w[2] + 1)), dtype=torch.int32, device=s.device)
        cell_of, order, cell_start, pos_sorted = self._cell_of, self._order, self._cell_start, self._pos_sorted
        if (self.domain is not None and getattr(self.domain, "fixed_capacity", False) and self.n_builds > 0 and not self.sort_particles
                and self._ref is not None and self._ref.shape[0] == s.N and self.nlist is not None and self.nlist.numel() == s.N * self.pitch):
            # every rebuild after the first of a fixed-capacity system: nothing to size, nothing to read back -- binning, sorted
            # copy, range table, search and commit in six launches (htfs_rebuild_nlist_ghosts) where the separate calls take ten
            capturing = getattr(self, "_capturing", False)
            if not capturing:
                self._poll_row_overflow()   # the PREVIOUS build's largest row (pinned copy behind it): no wait
            # the binning scratch is this object's own and every completed build leaves its counts zero (cell_order_kernel): no memset
            clean = getattr(self, "_scratch_clean", None) == (self._bin_scratch.data_ptr(), ncell)
            self._scratch_clean = None   # (a call that fails half way leaves the counts dirty: the next one zeroes them again)
            check(lib.htfs_rebuild_nlist_ghosts(s.pos.data_ptr(), s.scalar_code, s.N, Ntot, C.byref(sbox), self.r_list, C.byref(n3), C.byref(w3),
                                                cell_of.data_ptr(), self._bin_scratch.data_ptr(), cell_start.data_ptr(), order.data_ptr(),
                                                pos_sorted.data_ptr(), self.pitch, int(self.type_split), self.n_neigh.data_ptr(),
                                                self.head_list.data_ptr(), self.nlist.data_ptr(), self._max.data_ptr(), self._ref.data_ptr(),
                                                None, self._ranges.data_ptr(), int(clean), self._image_lengths(), stream))
            self._scratch_clean = (self._bin_scratch.data_ptr(), ncell)
            if getattr(self, "_max_host", None) is None:
                self._max_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            if not (capturing and getattr(self, "_mirrored", False)):   # (BrickRun: the next check kernel carries it to the host)
                self._max_host.copy_(self._max, non_blocking=True)
            if not capturing:
                self._max_event = torch.cuda.Event()
                self._max_event.record(torch.cuda.current_stream(s.device))
            if self._rule is not None:
                self._rule.reset()
                self._dd_prev = None
            self._grid = (n3, w3, ncell)
            self.n_builds += 1
            return
        check(lib.htfs_cell_index(s.pos.data_ptr(), s.scalar_code, Ntot, C.byref(sbox), C.byref(n3),
                                  cell_of.data_ptr(), stream))
        if self.sort_particles and s.N > 0:
            key = cell_of[: s.N].to(torch.int64)
            if self.domain is not None and self.domain.world > 1:
                # the slab layout [interior | left | both | right] (domain.py) carries the halo slices and
                # the interior row range: renumber inside each class only
                key = key + self.domain.row_classes() * int(n[0] * n[1] * n[2])
            perm = torch.sort(key, stable=True)[1]
            s.pos[: s.N] = s.pos[: s.N].index_select(0, perm)
            s.vel = s.vel.index_select(0, perm)
            cell_of[: s.N] = cell_of[: s.N].index_select(0, perm)
        # bin the particles: counting sort by cell, ascending index inside a cell (deterministic)
        check(lib.htfs_cell_sort(cell_of.data_ptr(), Ntot, ncell, self._bin_scratch.data_ptr(), cell_start.data_ptr(),
                                 order.data_ptr(), stream))
        # cell members contiguous: coalesced candidate reads
        fixed = self.domain is not None and getattr(self.domain, "fixed_capacity", False)
        if fixed:
            # fixed-capacity arrays (brick.py): inert rows are in no cell -- only the binned particles have an entry in ``order``
            check(lib.htfs_gather4_tagged_live(pos_sorted.data_ptr(), s.pos.data_ptr(), order.data_ptr(), s.scalar_code, Ntot,
                                               cell_start.data_ptr() + 4 * ncell, int(self.type_split), stream))
        else:
            check(lib.htfs_gather4_tagged(pos_sorted.data_ptr(), s.pos.data_ptr(), order.data_ptr(), s.scalar_code, Ntot,
                                          int(self.type_split), stream))
        if self.pitch is None:
            # a sphere of r_list at the mean density, with generous head-room
            L = s.box3x3[1] - s.box3x3[0]
            n_real = getattr(self.domain, "n_global", None) if fixed else None   # (capacity rows are not particles)
            rho = (n_real if n_real else Ntot) / float(np.prod(L))
            dims = int(np.sum(n > 1)) or 3
            est = rho * (4.0 / 3.0 * math.pi * self.r_list ** 3 if dims == 3 else math.pi * self.r_list ** 2 * L[2])
            self.pitch = max(8, int(math.ceil(est * 1.5 / 8.0)) * 8)
        capturing = getattr(self, "_capturing", False)
        if fixed and self.n_builds > 0 and not capturing:
            self._poll_row_overflow()   # the PREVIOUS build's largest row (pinned copy behind it): no wait
        while True:
            if self.n_neigh is None or self.n_neigh.shape[0] != s.N:
                self.n_neigh = torch.zeros(s.N, dtype=torch.int32, device=s.device)
                self.head_list = torch.zeros(s.N, dtype=torch.int32, device=s.device)
                if fixed:
                    self.domain.attach_n_neigh(self.n_neigh)   # a rebuild empties the rows that became inert
            if self.nlist is None or self.nlist.numel() != s.N * self.pitch:
                self.nlist = torch.empty(s.N * self.pitch, dtype=torch.int32, device=s.device)
            check(lib.htfs_build_nlist(s.pos.data_ptr(), pos_sorted.data_ptr(), s.scalar_code, s.N, Ntot, C.byref(sbox), self.r_list,
                                       C.byref(n3), C.byref(w3), cell_start.data_ptr(), self.pitch, int(self.type_split),
                                       self.n_neigh.data_ptr(), self.head_list.data_ptr(), self.nlist.data_ptr(),
                                       self._max.data_ptr(), self._ranges.data_ptr(), stream))
            if fixed and self.n_builds > 0:
                # no read-back in a rebuild of a fixed-capacity system: an overflowing row is reported one build late
                if getattr(self, "_max_host", None) is None:
                    self._max_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._max_host.copy_(self._max, non_blocking=True)
                if not capturing:
                    self._max_event = torch.cuda.Event()
                    self._max_event.record(torch.cuda.current_stream(s.device))
                break
            mx = int(self._max.item())
            if mx <= self.pitch:
                break
            self.pitch = int(math.ceil(mx * 1.2 / 8.0)) * 8
        if self._ref is None or self._ref.shape[0] != s.N:
            self._ref = s.pos[: s.N].clone()
        else:
            self._ref.copy_(s.pos[: s.N])
        if self._rule is not None:  # displacements are measured from the new reference positions
            self._rule.reset()
            self._dd_prev = None
        self._grid = (n3, w3, ncell)
        self.n_builds += 1

    # ------------------------------------------------------------------ device-side decision
    def _device_ok(self):
        return (self.device_decision and self._ref is not None and not self.sort_particles
                and (self.domain is None or self.domain.world == 1) and self.sys.n_ghost == 0
                and self._ref.shape[0] == self.sys.N and getattr(self, "_scr_n", (None,))[0] == self.sys.N)

    def _poll_overflow(self):
        """The previous check's (largest row, rebuild count), copied to pinned host memory behind it."""
        if self._stat_event is not None:
            self._stat_event.synchronize()  # recorded a whole check period ago: long since complete
            self._stat_event = None
            if int(self._stat_host[0]) > self.pitch:
                raise RuntimeError("neighbor list row overflow (%d entries, pitch %d) in a device-decided rebuild: "
                                   "construct CellNlist with a larger pitch" % (int(self._stat_host[0]), self.pitch))

    def _poll_row_overflow(self):
        ev = getattr(self, "_max_event", None)
        if ev is not None:
            ev.synchronize()
            self._max_event = None
            if int(self._max_host[0]) > self.pitch:
                raise RuntimeError("neighbor list row overflow (%d entries, pitch %d) in a rebuild without read-back: construct "
                                   "CellNlist with a larger pitch" % (int(self._max_host[0]), self.pitch))

    def check_and_rebuild_on_device(self):
        """NeighborList::compute at a check step with the decision left to the device: distance check, then
        the whole rebuild gated on its result.  No host synchronisation."""
        s = self.sys
        capturing = getattr(self, "_capturing", False)  # inside Simulation's hipGraph capture: no host-side waits
        if not capturing:
            self._poll_overflow()
        if self._stat is None:
            self._stat = torch.zeros(2, dtype=torch.int32, device=s.device)  # [largest row of the last rebuild, rebuilds]
            self._stat_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        n3, w3, ncell = self._grid
        stream = C.c_void_p(raw_stream(s.device.index))
        # distance check, gate, index + sort + sorted copy + search + commit behind it, status words to pinned memory: one call
        # (htf_standin.h htfs_check_rebuild_nlist -- a dozen Python-level calls and torch ops before)
        check(lib.htfs_check_rebuild_nlist(s.pos.data_ptr(), s.scalar_code, s.N, C.byref(s.box), self.r_list, C.byref(n3), C.byref(w3),
                                           self._cell_of.data_ptr(), self._bin_scratch.data_ptr(), self._cell_start.data_ptr(),
                                           self._order.data_ptr(), self._pos_sorted.data_ptr(), self.pitch, int(self.type_split),
                                           self.n_neigh.data_ptr(), self.head_list.data_ptr(), self.nlist.data_ptr(),
                                           self._stat.data_ptr(), self._ref.data_ptr(), self._disp.data_ptr(),
                                           (self.r_buff / 2.0) ** 2, self._stat_host.data_ptr(), self._ranges.data_ptr(), stream))
        if not capturing:
            self.mark_check_enqueued()

    def mark_check_enqueued(self):
        """An event behind the check's read-back; _poll_overflow waits for it one check later."""
        self._stat_event = torch.cuda.Event()
        # (an explicit device index: without one torch asks the runtime for the device count on every call, ~8 us)
        idx = self.sys.device.index
        self._stat_event.record(torch.cuda.current_stream(idx if idx is not None else torch.cuda.current_device()))

    def graph_key(self):
        """Addresses of every buffer a captured check + rebuild carries by value (Simulation._run_graphed)."""
        bufs = (self.n_neigh, self.head_list, self.nlist, self._ref, self._disp, self._stat, self._stat_host,
                getattr(self, "_cell_of", None), getattr(self, "_order", None), getattr(self, "_cell_start", None),
                getattr(self, "_bin_scratch", None), getattr(self, "_pos_sorted", None), getattr(self, "_ranges", None))
        return tuple(0 if b is None else b.data_ptr() for b in bufs) + (int(self.type_split), self.check_period)

    def device_builds(self):
        """Rebuilds the device has decided on so far (synchronises; for reports, not for the step loop)."""
        return int(self._stat[1].item()) if self._stat is not None else 0

    def _deferred_needs_update(self):
        """One check of the deferred rule (see DeferredRebuildRule).  Asynchronous form: the displacement kernel, a 4-byte
        MAX all-reduce on the device, a copy to pinned memory and an event -- nothing waits; the PREVIOUS check's value is
        read (its event completed a whole check period ago) and the decision follows from it."""
        import torch.distributed as dist
        s = self.sys
        if self._rule is None:
            self._rule = DeferredRebuildRule(self.r_buff / 2.0)
            self._dd_dev = [torch.zeros(1, dtype=torch.float32, device=s.device) for _ in range(3)]
            self._dd_host = [torch.zeros(1, dtype=torch.float32).pin_memory() if s.device.type == "cuda"
                             else torch.zeros(1, dtype=torch.float32) for _ in range(3)]
        buf = self._dd_dev[self._dd_i % 3]
        buf.zero_()
        check(lib.htfs_max_displacement2(s.pos.data_ptr(), self._ref.data_ptr(), s.scalar_code, s.N,
                                         C.byref(s.box), buf.data_ptr(),
                                         C.c_void_p(raw_stream(s.device.index))))
        alone = self.domain.world == 1   # (a replica brick: nobody to agree with)
        if self.deferred_reference:
            # the host-decided twin: all-reduce and read NOW (a drained queue per check), same rule, same one-check lag
            if not alone:
                dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.domain.group)
            now = float(np.sqrt(max(float(buf.item()), 0.0)))
            if self._dd_prev is not None:
                self._rule.push(self._dd_prev)
            self._dd_prev = now
        else:
            if not alone and getattr(self.domain, "transport", None) == "peer":
                self.domain.allreduce_max(buf)   # one launch through the peers' tables: no library in the step (csrc/mailbox.hip)
            elif not alone:
                work = dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.domain.group, async_op=True)
                work.wait()  # (RCCL: the current stream waits for the collective; the host does not)
            host = self._dd_host[self._dd_i % 3]
            host.copy_(buf, non_blocking=True)
            ev = torch.cuda.Event()
            idx = s.device.index
            ev.record(torch.cuda.current_stream(idx if idx is not None else torch.cuda.current_device()))
            if self._dd_prev is not None:
                h_prev, ev_prev = self._dd_prev
                ev_prev.synchronize()  # recorded a whole check period ago
                self._rule.push(float(np.sqrt(max(float(h_prev[0]), 0.0))))
            self._dd_prev = (host, ev)
        self._dd_i += 1
        if self._rule.decide():
            self._rule.reset()
            self._dd_prev = None  # what was measured against the old reference positions dies with them
            return True
        return False

    @property
    def dangerous_builds(self):
        return self._rule.dangerous if self._rule is not None else 0

    def needs_update(self):
        if self._ref is None:
            return True
        if (self.domain is not None and (self.domain.world > 1 or getattr(self.domain, "replica", False))
                and (self.device_decision or self.deferred_reference)):
            return self._deferred_needs_update()
        s = self.sys
        self._disp.zero_()
        check(lib.htfs_max_displacement2(s.pos.data_ptr(), self._ref.data_ptr(), s.scalar_code, s.N,
                                         C.byref(s.box), self._disp.data_ptr(),
                                         C.c_void_p(raw_stream(s.device.index))))
        if self.domain is not None and self.domain.world > 1:
            # every rank must take the same rebuild decision (the rebuild communicates)
            import torch.distributed as dist
            dist.all_reduce(self._disp, op=dist.ReduceOp.MAX, group=self.domain.group)
        # the threshold in fp32, as the device-side gate compares it (htfs_set_gate)
        return float(self._disp.item()) > float(np.float32((self.r_buff / 2.0) ** 2))

    def compute(self, timestep):
        """NeighborList::compute(timestep): rebuild if the distance check trips.  Under domain
        decomposition a second compute sharing this list in the same step finds the step's
        check / halo already done (HOOMD's NeighborList caches per timestep too)."""
        if self.domain is not None and self._ref is not None and getattr(self, "_step_done", None) == timestep:
            return
        self._step_done = timestep
        if self._device_ok():
            if timestep % self.check_period == 0:
                self.check_and_rebuild_on_device()
            return
        if self._ref is None or (timestep % self.check_period == 0 and self.needs_update()):
            self.build()
        elif self.domain is not None:
            # per-step forward halo of ghost positions: posted here, awaited by the force compute
            # after its interior rows (Context.compute_forces_overlapped) or by finish_halo()
            self.domain.exchange_begin()

    def finish_halo(self):
        if self.domain is not None:
            self.domain.exchange_end()


class Group:
    """hoomd.group.tags(first, last) analogue: a contiguous tag range [first, first + count)."""

    def __init__(self, first, count):
        self.first, self.count = int(first), int(count)

    def __len__(self):
        return self.count


class NVE:
    """hoomd.md.integrate.nve analogue (leapfrog form, unit mass); ``group``: only those
    particles move (enable_mapped_nlist returns the all-atom group for this)."""

    def __init__(self, system, dt, group=None):
        self.sys = system
        self.dt = float(dt)
        self.group = group

    def randomize_velocities(self, kT, seed):
        self.sys.randomize_velocities(kT, seed)
        if self.group is not None:
            g = self.group
            self.sys.vel[:g.first, :3] = 0
            self.sys.vel[g.first + g.count:, :3] = 0
        return self

    def step(self):
        s = self.sys
        first, n = (0, s.N) if self.group is None else (self.group.first, self.group.count)
        esz = 4 * (8 if s.dtype == torch.float64 else 4)
        check(lib.htfs_nve_step(s.pos.data_ptr() + first * esz, s.vel.data_ptr() + first * esz,
                                s.force.data_ptr() + first * esz, s.scalar_code, n,
                                self.dt, C.byref(s.box), C.c_void_p(raw_stream(s.device.index))))


class FusedStep:
    """One MD step as ONE launch (round 6): the integrator -- and, under a BrickDomain, the pack of the next step's halo messages --
    as the EPILOGUE of the one-kernel force step (include/htf_standin.h htfs_step_epilogue).  A finished row's lanes go on with
    v += f dt and x(t + dt) = wrap(x + v dt), the latter written into the OTHER position array (every wave still reads x(t) of
    everybody): ``System.pos`` and a twin are swapped behind every launch.  Same bits as force kernel + htfs_nve_step
    (+ htfs_brick_nve_halo): tests/test_gpu_standin.py, test_gpu_brick.py.

    ``available``: whether the context honours the epilogue (a built-in closed form on the one-kernel route, no virial, no batching,
    plain NVE of every particle, transport other than "peer"); when not, ``step()`` is the classic force launch + integrator launch.
    ``arrays(pos)``: the htf_hoomd_arrays of the CURRENT position array (one per twin, rebuilt when the list's buffers change)."""

    def __init__(self, system, nlist, ctx, nve):
        self.sys, self.nl, self.ctx, self.nve = system, nlist, ctx, nve
        self.dom = nlist.domain
        self._twin = None
        self._slot_of = {}
        self._arr = {}
        self._arr_key = None
        self.available = self._register()

    def _register(self):
        s, dom, ctx = self.sys, self.dom, self.ctx
        if os.environ.get("HTF_NO_STEP_EPILOGUE") == "1" or self.nve.group is not None or not s.pos.is_cuda:
            return False
        fixed = dom is not None and getattr(dom, "fixed_capacity", False)
        if dom is not None and (not fixed or not dom.kernels or dom.transport == "peer"):
            return False
        self._twin = torch.full_like(s.pos, float("nan"))
        self._twin[:, 3] = s.pos[:, 3]
        ok = True
        for slot, (cur, nxt) in enumerate(((s.pos, self._twin), (self._twin, s.pos))):
            kw = {}
            if fixed:
                dom.enable_row_slots()
                direct = dom.transport == "local"
                kw = dict(brick=dom.geom, row_slots=dom.row_slots, halo_send=None if direct else dom.halo_send,
                          ghost_direct=nxt[dom.cap:] if direct else None)
            ok = ctx.set_step_epilogue(slot, s.vel, nxt, self.nve.dt, s.box, **kw) and ok
            self._slot_of[cur.data_ptr()] = slot
        if not ok:
            self._twin = None
            return False
        self._home = s.pos.data_ptr()    # the array every check period starts and ends on (captured cycles carry it by value)
        if fixed and dom.transport == "local":
            # a rank that is its own neighbor: the epilogue writes the LIVE rows of a message straight into the other array's ghost
            # region; the inert rows behind a message's count are the re-plan's business -- in both arrays
            dom.after_replan.append(lambda: self._twin[dom.cap:, :3].fill_(float("nan")))
        return True

    def arrays(self):
        s, nl = self.sys, self.nl
        key = (nl.n_neigh.data_ptr(), nl.head_list.data_ptr(), nl.nlist.data_ptr(), s.force.data_ptr(), s.N)
        if key != self._arr_key:
            self._arr_key, self._arr = key, {}
        a = self._arr.get(s.pos.data_ptr())
        if a is None:
            a = self._arr[s.pos.data_ptr()] = self.ctx.make_arrays(s.pos, s.N, nl.n_neigh, nl.head_list, nl.nlist, s.box, s.force)
        return a

    def forces_and_integrate(self, ts):
        """The force rows of step ``ts`` (interior rows while a posted halo is in flight, boundary rows behind it) and the
        integrator: ONE launch each way, the positions swapped behind it -- except on the last step of a check period that finds
        the positions in their home array (an odd period: four swaps and one classic step), so that every period starts on the
        same array whatever the period and wherever the run began: a captured period replays from fixed addresses."""
        s, dom, ctx = self.sys, self.dom, self.ctx
        if dom is not None and dom.pending and (not dom.overlaps or 20 * dom.n_interior < s.N):
            dom.exchange_end()          # nothing in flight to hide, or next to no interior rows to hide it behind: one launch
        P = self.nl.check_period
        fused = self.available and not (ts % P == P - 1 and s.pos.data_ptr() == self._home)
        if fused:
            ctx.use_step_epilogue(self._slot_of[s.pos.data_ptr()])
            try:
                ctx.compute_forces_overlapped(ts, self.arrays(), dom)
            finally:
                ctx.use_step_epilogue(None)
            nxt = self._twin
            self._twin, s.pos = s.pos, nxt          # x(t + dt) is in the other array now
            if dom is not None:
                dom._packed = True                   # (the boundary rows wrote the next step's halo messages)
            return
        ctx.compute_forces_overlapped(ts, self.arrays(), dom)
        if dom is not None and getattr(dom, "kernels", False) and self.nve.group is None:
            dom.nve_step(self.nve.dt)               # integrator + next step's halo messages, one launch
        else:
            self.nve.step()

    def step(self, ts):
        """NeighborList::compute + the step (what bench.py's loop does on one GPU)."""
        self.nl.compute(ts)
        self.forces_and_integrate(ts)


# --------------------------------------------------------------------------- run loop
_current = {"sim": None}


def current_simulation():
    """hoomd.context.current analogue."""
    return _current["sim"]


class Simulation:
    """System::run analogue: per step, every attached force computes (ForceCompute::
    compute), the net force is formed, the integrator advances, computes observe."""

    def __init__(self, system):
        self.system = system
        self.forces = []     # hoomd.context.current.forces
        self.computes = []   # system.addCompute(...) without forces
        self.integrator = None
        _current["sim"] = self

    def nlist_cell(self, r_buff=0.4, check_period=1, pitch=None, device_decision=True):
        """hoomd.md.nlist.cell(): r_cut comes from the subscribers (nlist.subscribe).  The rebuild decision stays
        on the device (CellNlist.device_decision) wherever the list qualifies: one rank, no particle sorter."""
        return CellNlist(self.system, r_cut=0.0, r_buff=r_buff, pitch=pitch, check_period=check_period,
                         device_decision=device_decision)

    def integrate_nve(self, dt, group=None):
        self.integrator = NVE(self.system, dt, group=group)
        return self.integrator

    @property
    def net_force(self):
        return self.system.force

    def compute_forces(self):
        s = self.system
        ts = s.timestep
        for f in self.forces:
            f.compute(ts)
        if len(self.forces) == 1:
            s.force = self.forces[0].force
        elif self.forces:
            s.force = self.forces[0].force.clone()
            for f in self.forces[1:]:
                ops.add_scalar4(s.force, f.force)
        for c in self.computes:
            c.compute(ts)

    def _graph_cycle(self):
        """Steps per replayable cycle, or 0: one force compute that says its step is a fixed launch sequence
        (tfcompute.graph_safe), whose neighbor list decides its rebuilds on the device, plain NVE, nothing observing."""
        if len(self.forces) != 1 or self.computes or not isinstance(self.integrator, NVE) or self.integrator.group is not None:
            return 0
        f = self.forces[0]
        nl = getattr(f, "_nlist", None)
        if not getattr(f, "graph_safe", lambda: False)() or not isinstance(nl, CellNlist) or not nl._device_ok() or nl._stat is None:
            return 0
        return nl.check_period

    def _run_graphed(self, nsteps, soft=False):
        """Launch-bound systems (a few thousand particles: a step is ~15 kernels of ~1 us behind ~70 us of host
        enqueue): one check period of steps is captured into a hipGraph once and replayed, one launch per cycle.
        Every kernel reads its decisions from the device (gated rebuild), so the captured sequence is the step."""
        s = self.system
        cycle = self._graph_cycle()
        if cycle == 0 or nsteps < 4 * cycle:
            return nsteps
        while s.timestep % cycle != 0 and nsteps > 0:  # cycles start at a check step
            self._step()
            nsteps -= 1
        f, nl = self.forces[0], self.forces[0]._nlist
        # (everything a captured launch carries by value: a change of any of it re-captures -- sizes, scalars, and the ADDRESS of
        #  every buffer the step touches: rebinding sysm.pos / vel, a list whose buffers were re-allocated)
        key = (id(f), id(nl), id(self.integrator), cycle, s.N, nl.pitch, f.graph_key(), float(self.integrator.dt), float(nl.r_list),
               float(nl.r_buff), tuple(float(x) for x in np.asarray(s.box3x3).ravel()), s.pos.data_ptr(), s.vel.data_ptr(),
               s.force.data_ptr(), nl.graph_key())
        if getattr(self, "_graph_key", None) != key:
            torch.cuda.synchronize()
            nl._poll_overflow()
            g = torch.cuda.CUDAGraph()
            ts0 = s.timestep
            host_counters = (getattr(f, "_calls", 0), getattr(nl, "_step_done", None))  # what recording the cycle advances
            nl._capturing = True
            try:
                with torch.cuda.graph(g):
                    for _ in range(cycle):
                        self._step()
            except RuntimeError:
                if not soft:
                    raise
                # soft (run() chose the replay by itself): something in this step does not capture -- an op that synchronises,
                # say.  Step by step then, and no second attempt.
                self._no_graph, self._graph, self._graph_key = True, None, None
                nl._capturing = False
                s.timestep = ts0
                f._calls, nl._step_done = host_counters  # the aborted recording ran no step
                torch.cuda.synchronize()
                return nsteps
            finally:
                nl._capturing = False
                s.timestep = ts0  # a capture records, it does not run
            f._calls, nl._step_done = host_counters  # (the replay loop below counts the steps that really run)
            self._graph, self._graph_key = g, key
        # The neighbor-row overflow report (a pinned copy inside the captured cycle) is polled EVERY cycle, on the event
        # recorded two replays earlier -- long since complete, so the host still runs two cycles ahead of the device --
        # instead of every 8th cycle: at most 3 check periods run on a truncated list before the RuntimeError.
        # (an event per replay costs ~4 us of host time: with a one-step cycle -- check_period 1 -- that is every step, so
        #  the poll runs every `every`-th replay, i.e. at least every 4 steps)
        behind = []
        every = max(1, -(-4 // cycle))
        n_replay = 0
        while nsteps >= cycle:
            if n_replay % every == 0 and len(behind) >= 2:
                nl._stat_event = behind.pop(0)
                nl._poll_overflow()
            self._graph.replay()
            f._calls = getattr(f, "_calls", 0) + cycle
            s.timestep += cycle
            nsteps -= cycle
            if n_replay % every == 0:
                nl.mark_check_enqueued()
                behind.append(nl._stat_event)
                nl._stat_event = None
            n_replay += 1
        for ev in behind:
            nl._stat_event = ev
            nl._poll_overflow()
        return nsteps

    def _step(self):
        self.compute_forces()
        if self.integrator is not None:
            self.integrator.step()
        self.system.timestep += 1

    def _choose_by_measurement(self, nsteps):
        """graph=None: which way is this step faster?  Until round 5 the replay was chosen for any run of >= 256 qualifying steps;
        at C3 size (a step of ~75 us, kernel-bound) the driver's line had the replay 8 % SLOWER than the stepwise loop, while at
        a few thousand particles (host-enqueue-bound) it is ~2x faster.  So the first long run measures both on its own first
        steps -- about 32 stepwise, then about 32 replayed (the capture itself is not timed) -- and keeps the faster until
        something the graph carries by value changes.  Same trajectory either way.  -> steps consumed."""
        import time
        cycle = self._graph_cycle()
        if cycle == 0:
            return 0
        s = self.system
        f, nl = self.forces[0], self.forces[0]._nlist
        n = max(1, 32 // cycle) * cycle
        if nsteps < 2 * n + 6 * cycle:
            return 0
        done = 0
        while s.timestep % cycle != 0:
            self._step()
            done += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            self._step()
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - t0) / n
        done += n
        left = self._run_graphed(4 * cycle, soft=True)     # capture (+ a few replays), untimed
        done += 4 * cycle
        if left or self._graph is None:                      # did not capture: stepwise it is
            for _ in range(left):
                self._step()
            self.graph_choice = {"use_graph": False, "why": "the step does not capture", "stepwise_us": t_step * 1e6, "key": None}
            return done
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self._run_graphed(n)
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / n
        done += n
        self.graph_choice = {"use_graph": t_graph < t_step, "stepwise_us": t_step * 1e6, "graph_us": t_graph * 1e6,
                             "steps_per_leg": n, "key": self._graph_key}
        return done

    def run(self, nsteps, graph=None):
        """graph=True (or HTF_RUN_GRAPH=1): replay whole check periods as one hipGraph launch where the step
        qualifies (_graph_cycle); anything else, and the remainder, runs step by step.  graph=None (the default): for runs of at
        least 256 steps of a qualifying step the FASTER of the two, measured once on the run's own first steps
        (_choose_by_measurement; ``self.graph_choice`` records it) -- small systems are bound by the host's enqueue (the
        reference's own 256-particle benchmark: 18 k steps/s stepwise, 33 k replayed), at 131 072 particles the kernels bound the
        step and the stepwise loop is as fast or faster -- unless HTF_RUN_GRAPH=0; graph=False: always step by step."""
        nsteps = int(nsteps)
        env = os.environ.get("HTF_RUN_GRAPH")
        auto = graph is None and env != "1"
        if graph is None:
            graph = env == "1"
            if env not in ("0", "1") and nsteps >= 256 and self.system.pos.is_cuda and not getattr(self, "_no_graph", False):
                choice = getattr(self, "graph_choice", None)
                if choice is None or (choice["key"] is not None and choice["key"] != getattr(self, "_graph_key", None)):
                    nsteps -= self._choose_by_measurement(nsteps)
                    choice = getattr(self, "graph_choice", None)
                graph = bool(choice and choice["use_graph"])
        if graph:
            # (chosen by default, not asked for: a step that turns out not to capture must still run -- soft)
            nsteps = self._run_graphed(nsteps, soft=auto)
        for _ in range(nsteps):
            self._step()


class BrickRun:
    """The decomposed MD loop over a fixed-capacity ``brick.BrickDomain``: NeighborList::compute (distance check, migration +
    ghost plan + list rebuild when it trips, else the per-step ghost halo), the force rows that see no ghost while the halo is
    in flight, the rest behind it, the integrator.

    ``run(n)`` steps eagerly -- about a dozen launches per step from Python, which at 16 k rows per rank is the step's floor
    (27-32 us measured in round 4 against an 11 us kernel).  ``run(n, graph=True)`` replays whole check periods from TWO
    hipGraphs per rank, captured once: A = [check, period x (halo | interior rows | boundary rows | integrate)] and
    B = [check, migrate, re-plan, list rebuild, period x step].  Nothing in either is sized by a count the host knows (fixed
    capacities, inert rows, device-resident counts), the halo and the migration messages are RCCL calls on the captured streams
    (csrc/halo.hip) -- or, in replica mode with transport "local", written by the pack kernel itself.  Which graph runs next is
    decided WITHOUT draining the queue: every cycle's check leaves [largest displacement^2, cycle number] in pinned memory, the
    host reads the word of the cycle it launched last -- available as soon as the device has STARTED that cycle, i.e. a whole
    cycle of device work before it is needed -- and feeds DeferredRebuildRule, the rule of the eager no-read-back path: same
    decisions, same steps rebuilt, bit-identical trajectory (tests/test_gpu_brick.py)."""

    def __init__(self, system, nlist, ctx, integrator):
        self.sys, self.nl, self.ctx, self.nve = system, nlist, ctx, integrator
        self.dom = nlist.domain
        if self.dom is None or not getattr(self.dom, "fixed_capacity", False):
            raise ValueError("BrickRun drives a CellNlist whose domain is a BrickDomain")
        self._graphs = None
        self._mirror = None
        self.n_rebuild_cycles = 0
        self.n_cycles = 0
        self._fstep = None

    # ------------------------------------------------------------------ pieces of a step
    @property
    def fstep(self):
        """The step as one launch where the context honours it (FusedStep: integrator and halo pack as the force kernel's epilogue,
        positions ping-ponging between two arrays); made on first use -- the list must have been built."""
        if self._fstep is None:
            self._fstep = FusedStep(self.sys, self.nl, self.ctx, self.nve)
        return self._fstep

    def _arrays(self):
        return self.fstep.arrays()     # (the htf_hoomd_arrays of the CURRENT position array and list buffers)

    def _integrate(self):
        if self.dom.kernels and self.nve.group is None:
            self.dom.nve_step(self.nve.dt)   # integrator + next step's halo messages, one launch
        else:
            self.nve.step()

    def _force_rows(self, ts):
        """The forces of step ``ts`` alone (no integration): interior rows | halo | boundary rows, or one launch."""
        if self.dom.pending and (not self.dom.overlaps or 20 * self.dom.n_interior < self.sys.N):
            self.dom.exchange_end()          # nothing in flight to hide, or next to no interior rows to hide it behind: one launch
        self.ctx.compute_forces_overlapped(ts, self._arrays(), self.dom)

    def advance(self, ts):
        """Forces of step ``ts`` + the integrator + the next step's halo messages: one launch (FusedStep) or the three pieces."""
        self.fstep.forces_and_integrate(ts)

    def step(self):
        """One eager step (what bench.py's loop does)."""
        s = self.sys
        ts = s.timestep
        self.nl.compute(ts)
        self.advance(ts)
        s.timestep += 1

    # ------------------------------------------------------------------ graph replay
    def _check_kernels(self):
        """The distance check of a cycle, on the device: largest displacement^2 since the last rebuild, all-reduced, and the cycle
        counter, copied to pinned memory."""
        s, nl = self.sys, self.nl
        # [d2, cycle] <- [largest displacement^2, cycle + 1] in one launch (the accumulator and its reset live in _stat_work); the
        # same launch carries the status words of the cycle before -- the decomposition's counts and flags, the list's largest row
        # -- and, on one rank, its own result into pinned memory: no copy node in the chain
        alone = self.dom.world == 1
        check(lib.htfs_check_displacement2(s.pos.data_ptr(), nl._ref.data_ptr(), s.scalar_code, s.N, C.byref(s.box),
                                           self._stat_work.data_ptr(), self._stat.data_ptr(),
                                           self._stat_host.data_ptr() if (alone and self._mirror is not None) else None,
                                           C.byref(self._mirror) if self._mirror is not None else None,
                                           C.c_void_p(raw_stream(s.device.index))))
        if not alone:
            self.dom.allreduce_max(self._stat[0:1])     # RCCL on the captured stream ("native"), or one launch through the peers' tables ("peer")
        if not alone or self._mirror is None:
            self._stat_host.copy_(self._stat, non_blocking=True)

    def _cycle(self, rebuild):
        s, nl, P = self.sys, self.nl, self.nl.check_period
        self._check_kernels()
        if rebuild:
            # (capture: the kind of rebuild this graph holds is named, not taken from the domain's counter)
            self.dom._next_light = (rebuild == "light") if getattr(self.dom, "_capturing", False) else None
            try:
                nl.build()           # BrickDomain.rebuild (migration, classes, plan, halo -- or the halo alone) + the list
            finally:
                self.dom._next_light = None
        for i in range(P):
            if not (rebuild and i == 0):
                self.dom.exchange_begin()
            self.advance(s.timestep)
            s.timestep += 1

    def _capture(self):
        s, nl, dom = self.sys, self.nl, self.dom
        if not dom.kernels or dom.transport not in ("native", "local", "peer"):
            raise ValueError("graph replay needs the kernels backend and the 'native' (RCCL inside the capture), 'peer' or 'local' transport")
        if dom.world > 1 and dom._native is None and dom.transport != "peer":
            raise ValueError("graph replay over real ranks needs a device-side all-reduce of the distance check and capturable migration "
                             "messages: transport='native' (the library's own RCCL communicator) or 'peer' (no library at all)")
        if nl.n_builds < 2:
            raise RuntimeError("run a few eager steps through a rebuild first (RCCL connects and pinned buffers are made outside a capture)")
        dom.exchange_end()
        if self.fstep.available and s.pos.data_ptr() != self.fstep._home:
            raise RuntimeError("a captured cycle must start on the positions' home array (BrickRun.run steps eagerly to a check step first)")
        self._stat = torch.zeros(2, dtype=torch.float32, device=s.device)
        self._stat_work = torch.zeros(2, dtype=torch.int32, device=s.device)
        # the status words the host reads one cycle late travel with the check kernel (HTF_BRICK_MIRROR=0: a copy node each, as before)
        self._mirror = None
        if os.environ.get("HTF_BRICK_MIRROR", "1") != "0":
            if getattr(nl, "_max_host", None) is None:
                nl._max_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            m = _lib.Mirror()
            m.src[0], m.dst[0], m.words[0] = dom.counts.data_ptr(), dom._flags_host.data_ptr(), _lib.BC_WORDS
            m.src[1], m.dst[1], m.words[1] = nl._max.data_ptr(), nl._max_host.data_ptr(), 1
            m.n = 2
            self._mirror = m
        nl._mirrored = dom._mirrored = self._mirror is not None
        self._stat_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        self._stat_host_words = self._stat_host.view(torch.int32)
        self._rule = DeferredRebuildRule(nl.r_buff / 2.0)
        torch.cuda.synchronize()
        graphs = {}
        ts0, builds0, rebuilds0 = s.timestep, nl.n_builds, dom.n_rebuilds
        index0, light0 = dom._rebuild_index, dom.n_light
        nl._capturing = dom._capturing = True
        try:
            # A = [check, P steps]; B = [check, migrate + re-plan + list rebuild, P steps]; with BrickDomain(replan_every > 1) also
            # "light" = [check, list rebuild on the rows and messages as they are, P steps]
            for rebuild in ((False, True, "light") if dom.replan_every > 1 else (False, True)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._cycle(rebuild)
                graphs[rebuild] = g
                s.timestep = ts0       # a capture records, it does not run
        finally:
            nl._capturing = dom._capturing = False
            s.timestep, nl.n_builds, dom.n_rebuilds = ts0, builds0, rebuilds0
            dom._rebuild_index, dom.n_light = index0, light0
            nl._step_done = None
        self._graphs = graphs
        self._launched = 0          # cycle number of the last launched graph (the device counter counts from 1)
        self._read = 0              # cycle number whose check the rule has seen
        self._discard = set()       # cycles whose check was measured against reference positions that a rebuild then replaced
        self._rebuilt_at = []

    def _read_check(self, cycle):
        """Wait until the device has started ``cycle`` (its check has landed in pinned memory), then validate everything that
        finished before it."""
        import time
        h = self._stat_host
        hw = self._stat_host_words          # the same two words as integers: [bits of d2, cycle number (unsigned, wraps at 2^32)]
        spins, t0 = 0, None
        # the device counter is compared modulo 2^32 (it is never more than one cycle behind what was launched)
        while ((int(hw[1]) - cycle) & 0xFFFFFFFF) >= 0x80000000:
            spins += 1
            if spins % 4096 == 0:   # (a stuck device must not hold the host forever: HTF_BRICK_WAIT_S seconds, default 30)
                t0 = t0 or time.monotonic()
                if time.monotonic() - t0 > float(os.environ.get("HTF_BRICK_WAIT_S", "30")):
                    raise RuntimeError("the device never reached cycle %d (last seen %d)" % (cycle, int(hw[1]) & 0xFFFFFFFF))
        d2 = float(h[0])
        # cycle - 1 is complete: what its rebuild (if any) reported
        self.dom._raise_flags(int(self.dom._flags_host[_lib.BC_FLAGS]))
        mh = getattr(self.nl, "_max_host", None)
        if mh is not None and int(mh[0]) > self.nl.pitch:
            raise RuntimeError("neighbor list row overflow (%d entries, pitch %d) in a replayed rebuild" % (int(mh[0]), self.nl.pitch))
        return d2

    def run(self, nsteps, graph=False):
        nsteps = int(nsteps)
        P = self.nl.check_period
        if not graph:
            for _ in range(nsteps):
                self.step()
            return
        s = self.sys
        while s.timestep % P != 0 and nsteps > 0:
            self.step()
            nsteps -= 1
        if self._graphs is None:
            self._capture()
        rule = self._rule
        while nsteps >= P:
            if self._launched > self._read:
                # (at most one cycle is ever unread: the one launched last)
                d2 = self._read_check(self._launched)
                self._read = self._launched
                if self._read not in self._discard:
                    rule.push(float(np.sqrt(max(d2, 0.0))))
                else:
                    self._discard.discard(self._read)     # (read once: nothing older than _read is ever looked up again)
            rebuild = rule.decide()
            if rebuild:
                rule.reset()
                self._discard.add(self._launched + 1)   # that cycle's check runs BEFORE its rebuild: measured against the old reference
                self.n_rebuild_cycles += 1
                self._rebuilt_at.append(s.timestep)
                if len(self._rebuilt_at) > 4096:           # (a report for tests and tools, not a log of a production-length run)
                    del self._rebuilt_at[:2048]
                self.nl.n_builds += 1
                self.dom.n_rebuilds += 1
                # which rebuild: the domain's own rule (every replan_every-th re-plans), kept in step with the eager path
                if self.dom._rebuild_index % self.dom.replan_every != 0:
                    rebuild = "light"
                    self.dom.n_light += 1
                self.dom._rebuild_index += 1
            self._graphs[rebuild].replay()
            self._launched += 1
            self.n_cycles += 1
            s.timestep += P
            nsteps -= P
        if self._mirror is not None and self._launched > self._read:
            # the last cycle's status words have no later check kernel to carry them: two plain copies behind the last graph
            self.dom._flags_host.copy_(self.dom.counts, non_blocking=True)
            self.nl._max_host.copy_(self.nl._max, non_blocking=True)
        for _ in range(nsteps):
            self.step()

    @property
    def dangerous_builds(self):
        return self._rule.dangerous if getattr(self, "_rule", None) is not None else self.nl.dangerous_builds
