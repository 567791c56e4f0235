"""Particle domain decomposition for the force path: 1-D slabs along x, one rank per GPU,
forward ghost-POSITION halo over ``torch.distributed`` P2P (RCCL send/recv over xGMI on
the GPU box, gloo in the CPU tests).

What the reference inherits from HOOMD's ``Communicator``: particle
migration + ghost exchange when the neighbor list is rebuilt, and a ghost position update
every step.  Because F_i is a pure row reduction over particle i's own neighbor slots
(simmodel.py:542-555) there is NO reverse/force communication and no all-reduce in
inference: the only per-step message is ~0.5 MB of float4 positions per slab face.

Layout after ``rebuild()`` (HOOMD's): ``pos[:N]`` local particles, ``pos[N:N+n_ghost]``
ghosts = [from left neighbor | from right neighbor].  Ghosts keep the owner's raw
coordinates; the pair-vector build applies the minimum image of the GLOBAL box.
Local particles are ordered [interior | near the left face only | near BOTH faces | near the
right face only] ("near" = within r_ghost): the two halo messages are contiguous -- and, in a
slab thinner than 2 r_ghost, overlapping -- slices of ``pos`` (sent in place, no packing
kernel), and rows ``[0, n_interior)`` have no ghost in their neighbor lists, so their forces
can be evaluated while the halo is in flight (``exchange_begin`` / ``exchange_end``).
A slab may be as thin as r_ghost (then every ghost still comes from an adjacent slab); with two
ranks the two faces lead to the same peer, so there the slabs must be 2 r_ghost thick or a
particle would arrive twice.  The 131 072-particle box of the headline metric (L = 53.75,
r_ghost = 3.4) decomposes over 8 ranks this way (slab 6.72).

Message naming: "L>" = sent to my left neighbor (my particles within r_ghost of my left
face), "R>" = sent to my right neighbor.  Every rank posts sends in the order [L>, R>] and
receives in the order [L> from right, R> from left], which keeps grouped NCCL send/recv
pairs matched even when left == right (world size 2).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

TAG_L, TAG_R = 11, 12


class _StagedHalo:
    """gloo transport for device buffers: bounce through host copies.  Only the 2-ranks-on-one-GPU
    test rig uses it; on the GPU box the backend is nccl (= RCCL) and messages go device to
    device over xGMI."""

    def __init__(self, ops):
        self._back = []
        staged = []
        for op in ops:
            host = op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            if op.op is dist.irecv:
                self._back.append((op.tensor, host))
            staged.append(dist.P2POp(op.op, host, op.peer, op.group, op.tag))
        self._works = dist.batch_isend_irecv(staged)

    def wait(self):
        for w in self._works:
            w.wait()
        for dev, host in self._back:
            dev.copy_(host)


def _lib_error():
    from ._lib import last_error
    return last_error()


class _NativeHalo:
    """The per-step halo through libhtf_amd.so's own RCCL communicator (csrc/halo.hip): one grouped
    ncclSend x2 / ncclRecv x2 on a dedicated stream, two events, no Python objects per message."""

    _shared = {}

    @classmethod
    def shared(cls, rank, world, group, solo=False):
        """One communicator per process and job shape, kept until the process ends: bring-up is a collective (and costs tens of
        milliseconds), and a second communicator made after the first was destroyed hung the first hipGraph capture that used it
        (tests/test_gpu_brick.py, round 5)."""
        key = (int(rank), int(world), id(group), bool(solo))
        if key not in cls._shared:
            cls._shared[key] = cls(rank, world, group, solo=solo)
        return cls._shared[key]

    def __init__(self, rank, world, group, solo=False):
        """``solo``: a communicator of one rank (brick.py's replica mode: the rank is its own neighbor), no process group needed."""
        import ctypes as C
        from ._lib import lib, check
        self._C, self._lib, self._check = C, lib, check
        # 128 bytes of id + one status byte: if rank 0 cannot obtain an id, every rank learns it from the SAME
        # broadcast and gives up together (no rank is left waiting in a collective)
        ident = torch.zeros(129, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            if lib.htf_halo_unique_id(buf) == 0:
                ident[:128] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
                ident[128] = 1
        if solo:
            pass
        elif dist.get_backend(group) == "nccl":  # the id travels through whatever channel the job has
            dev = ident.cuda()
            dist.broadcast(dev, src=0, group=group)
            ident = dev.cpu()
        else:
            dist.broadcast(ident, src=0, group=group)
        if int(ident[128]) != 1:
            raise RuntimeError("rank 0 could not obtain an RCCL unique id: " + _lib_error())
        ident = ident[:128].contiguous()
        self._h = C.c_void_p()
        raw = (C.c_char * 128).from_buffer_copy(bytes(ident.numpy().tobytes()))
        check(lib.htf_halo_create(raw, int(rank), int(world), C.byref(self._h)))

    def begin(self, pos, left, right, send_left, send_right, recv_left, recv_right):
        from . import ops
        s = self._C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
        self._check(self._lib.htf_halo_exchange_begin(
            self._h, pos.data_ptr(), ops._dt(pos), int(left), int(right), send_left[0], send_left[1] - send_left[0],
            send_right[0], send_right[1] - send_right[0], recv_left[0], recv_left[1] - recv_left[0],
            recv_right[0], recv_right[1] - recv_right[0], s))

    def exchange(self, send, recv, caps, offs, neighbors, opposite, overlap=True):
        """brick.py's grouped exchange: message m = rows [offs[m], +caps[m]) of ``send`` to neighbors[m], ascending m; the message
        from the neighbor at offset index j into rows [offs[j], +caps[j]) of ``recv``, descending j.  ``overlap``: on the halo
        stream (ended by wait()); else on the current stream."""
        C = self._C
        n = len(caps)
        row = send.shape[1] * send.element_size()
        key = (send.data_ptr(), recv.data_ptr(), overlap)
        plan = getattr(self, "_plans", None)
        if plan is None:
            plan = self._plans = {}
        if key not in plan:
            VP, SZ, IN = C.c_void_p * n, C.c_size_t * n, C.c_int * n
            order = list(reversed(range(n)))
            plan[key] = (VP(*[send.data_ptr() + offs[m] * row for m in range(n)]), SZ(*[caps[m] * row for m in range(n)]),
                         IN(*[neighbors[m] for m in range(n)]),
                         VP(*[recv.data_ptr() + offs[j] * row for j in order]), SZ(*[caps[j] * row for j in order]),
                         IN(*[neighbors[j] for j in order]))
        sp, sb, sr, rp, rb, rr = plan[key]
        s = C.c_void_p(torch.cuda.current_stream(send.device).cuda_stream)
        self._check(self._lib.htf_halo_exchange_n(self._h, n, sp, sb, sr, n, rp, rb, rr, s, 1 if overlap else 0))
        self._overlapped = overlap

    def allreduce_max(self, value):
        s = self._C.c_void_p(torch.cuda.current_stream(value.device).cuda_stream)
        self._check(self._lib.htf_halo_allreduce_max_f32(self._h, value.data_ptr(), int(value.numel()), s))

    def wait(self):
        # (called from exchange_end on whatever stream is current then)
        s = self._C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self._lib.htf_halo_exchange_end(self._h, s))

    def __del__(self):
        h = getattr(self, "_h", None)
        try:
            if h and self._lib is not None and self not in type(self)._shared.values():
                self._lib.htf_halo_destroy(h)
                self._h = None
        except TypeError:   # interpreter teardown: the binding is already gone
            pass


class SlabDomain:
    def __init__(self, system, rank, world, r_ghost, fractions=None, group=None, transport="torch"):
        """``transport``: "torch" (default) = torch.distributed grouped P2P (RCCL underneath on the GPU box, gloo in
        the CPU tests); "native" = libhtf_amd.so's own RCCL communicator and halo stream; "auto" = native when the
        job runs on the nccl backend, the library could load librccl, and the first exchange reproduces what the
        torch transport delivers (checked once, at the first rebuild), else torch.  The native transport is OPT-IN
        until it has moved bytes between two real devices (tests/test_gpu_domain.py::test_rccl_halo_two_gpus skips
        on the 1-GPU boxes of this pool): ``HTF_HALO_TRANSPORT=auto|native`` selects it for bench.py."""
        self.sys = system
        self.transport_request = transport
        self.transport = "torch"
        self._native = None
        self.rank, self.world = int(rank), int(world)
        self.r_ghost = float(r_ghost)
        self.group = group
        lo, hi = float(system.box3x3[0][0]), float(system.box3x3[1][0])
        if fractions is None:
            cuts = np.linspace(0.0, 1.0, self.world + 1)
        else:  # comm.decomposition(x=[0.33]) style interior cut fractions
            cuts = np.concatenate([[0.0], np.asarray(fractions, dtype=np.float64), [1.0]])
            if len(cuts) != self.world + 1 or np.any(np.diff(cuts) <= 0):
                raise ValueError("need world-1 increasing cut fractions in (0, 1)")
        self.bounds = lo + cuts * (hi - lo)
        self.xlo, self.xhi = float(self.bounds[self.rank]), float(self.bounds[self.rank + 1])
        thinnest = float(min(np.diff(self.bounds)))
        if self.world == 2 and thinnest < 2.0 * self.r_ghost:
            raise ValueError("two slabs thinner than 2 * r_ghost: both faces lead to the same peer, a particle "
                             "near both would arrive there twice")
        if self.world > 2 and thinnest < self.r_ghost:
            raise ValueError("slab thinner than r_ghost: ghosts would have to come from beyond the adjacent slab")
        self.left = (self.rank - 1) % self.world
        self.right = (self.rank + 1) % self.world
        self.send_left = self.send_right = None   # (start, stop) row ranges of pos
        self.class_counts = (0, 0, 0, 0)          # rows: interior | left only | both | right only
        self.n_interior = 0
        self.n_from_left = self.n_from_right = 0
        self.n_migrated = 0
        self.local_counts = None                  # [world] particles per rank as of the last rebuild
        self._works = None

    @property
    def pending(self):
        """True between exchange_begin() and exchange_end()."""
        return self._works is not None

    # ------------------------------------------------------------------ helpers
    def _gather_counts(self, cnt):
        """cnt: small int64 device vector -> [world, len] host array.  One all_gather and ONE
        device->host copy: the only host synchronisation of a migration / ghost-plan phase."""
        out = [torch.empty_like(cnt) for _ in range(self.world)]
        dist.all_gather(out, cnt, group=self.group)
        return torch.stack(out).cpu().numpy()

    def _swap(self, to_left, to_right, n_from_right, n_from_left):
        """send [L>, R>], receive [L> from right, R> from left]; returns the two buffers."""
        width = to_left.shape[1:]
        from_right = torch.empty((n_from_right,) + tuple(width), dtype=to_left.dtype, device=to_left.device)
        from_left = torch.empty((n_from_left,) + tuple(width), dtype=to_left.dtype, device=to_left.device)
        self._swap_into(to_left, to_right, from_right, from_left)
        return from_right, from_left

    def _post(self, to_left, to_right, from_right, from_left):
        ops = []
        if to_left.numel():
            ops.append(dist.P2POp(dist.isend, to_left, self.left, self.group, TAG_L))
        if to_right.numel():
            ops.append(dist.P2POp(dist.isend, to_right, self.right, self.group, TAG_R))
        if from_right.numel():
            ops.append(dist.P2POp(dist.irecv, from_right, self.right, self.group, TAG_L))
        if from_left.numel():
            ops.append(dist.P2POp(dist.irecv, from_left, self.left, self.group, TAG_R))
        if not ops:
            return []
        if to_left.is_cuda and dist.get_backend(self.group) == "gloo":
            return [_StagedHalo(ops)]  # gloo moves host memory only (single-GPU test rigs)
        return dist.batch_isend_irecv(ops)

    def _swap_into(self, to_left, to_right, from_right, from_left):
        for w in self._post(to_left, to_right, from_right, from_left):
            w.wait()

    # ------------------------------------------------------------------ migration + ghost plan
    def rebuild(self):
        """Communicator::migrateParticles + exchangeGhosts: call before every neighbor-list
        build.  Afterwards system.N / pos / vel describe this rank's slab plus its ghosts."""
        s = self.sys
        if self.world == 1:
            s.n_ghost = 0
            return
        N = s.N
        pv = torch.cat([s.pos[:N], s.vel[:N]], dim=1)  # [N, 8]: a particle travels as one record
        # Destination and ghost class of every particle, and the stable (destination, class) order.
        # The integrator wraps into the global box, so the new owner is found by position and
        # is an adjacent slab by construction (rebuilds happen before anything moves r_buff/2).
        # Ghost class IN THE SLAB THE PARTICLE ENDS UP IN (the sender knows every slab's bounds): 0 interior,
        # 1 near the left face only, 2 near both faces (slabs thinner than 2 r_ghost), 3 near the right face only.
        # One stable sort on (destination, class) and ONE exchange of the 16 counts then carry both the migration
        # plan and the ghost plan: each rank can work out every rank's class counts after the migration, so the
        # second count exchange of the first version (and its device -> host round trip) is gone.
        kernels = pv.is_cuda and os.environ.get("HTF_DOMAIN_TORCH") != "1"
        if kernels:
            # on the device: one classification kernel + a stable counting sort over the 16 keys
            # (csrc/standin.hip: slab_classify_kernel, htfs_key_sort16) -- four launches where the torch
            # restatement below takes about twenty; the order and the counts are the same (tests/test_gpu_domain.py)
            cnt, order = self._classify_sort_device(N)
            pv = pv.index_select(0, order)
        else:
            x = pv[:, 0].contiguous()
            bnd = torch.as_tensor(self.bounds, dtype=x.dtype, device=x.device)
            owner = torch.bucketize(x, bnd[1:-1].contiguous(), right=True)
            if self.world == 2:
                dest = (owner != self.rank).to(torch.int64) * 2  # one peer: everything travels as the R> message
            else:
                go_left, go_right = owner == self.left, owner == self.right
                lost = (owner != self.rank) & ~go_left & ~go_right
                dest = go_left.to(torch.int64) + 2 * go_right.to(torch.int64) + 3 * lost.to(torch.int64)
            near_l, near_r = x < bnd[owner] + self.r_ghost, x >= bnd[owner + 1] - self.r_ghost
            cls = torch.where(near_l, torch.where(near_r, 2, 1), torch.where(near_r, 3, 0))
            key = dest * 4 + cls
            cnt = torch.zeros(16, dtype=torch.int64, device=x.device).index_add_(0, key, torch.ones_like(key))
            pv = pv.index_select(0, torch.sort(key, stable=True)[1])
        allc = self._gather_counts(cnt).reshape(self.world, 4, 4)  # [rank, destination, class]
        if allc[:, 3].any():
            raise RuntimeError("a particle crossed more than one slab between neighbor-list rebuilds")
        mine = allc[self.rank]
        n_stay, n_l, n_r = (int(mine[d].sum()) for d in range(3))
        pack_l = pv[n_stay:n_stay + n_l].contiguous()
        pack_r = pv[n_stay + n_l:n_stay + n_l + n_r].contiguous()
        in_r, in_l = allc[self.right][1], allc[self.left][2]  # class counts of what arrives from the right / left
        got_r, got_l = self._swap(pack_l, pack_r, int(in_r.sum()), int(in_l.sum()))
        self.n_migrated += int(got_r.shape[0] + got_l.shape[0])
        pv = torch.cat([pv[:n_stay], got_r, got_l], dim=0)
        N = int(pv.shape[0])
        # every segment arrives sorted by class: merging the three is 12 row-range copies whose sources and
        # destinations follow from the counts already on the host
        if kernels:
            pv = self._merge_segments_device(pv, mine[0], in_r, in_l)
        else:
            seg = torch.as_tensor(np.concatenate([mine[0], in_r, in_l]), dtype=torch.int64, device=pv.device)
            cls_all = torch.repeat_interleave(torch.arange(4, device=pv.device).repeat(3), seg)
            pv = pv.index_select(0, torch.sort(cls_all, stable=True)[1])
        new_pos, new_vel = pv[:, :4], pv[:, 4:]

        def after(q):  # rank q's class counts once everybody has migrated
            return allc[q][0] + allc[(q + 1) % self.world][1] + allc[(q - 1) % self.world][2]

        c0, c1, c2, c3 = (int(v) for v in after(self.rank))
        assert c0 + c1 + c2 + c3 == N
        # every rank's particle count after this migration, from the counts already on the host: what a data-parallel training
        # step needs to size its global batch without another collective or read-back (tfcompute._train_on_batch)
        self.local_counts = [int(after(q).sum()) for q in range(self.world)]
        self.class_counts = (c0, c1, c2, c3)
        self.n_interior = c0
        self.send_left = (c0, c0 + c1 + c2)
        self.send_right = (c0 + c1, N)
        cr, cl = after(self.right), after(self.left)
        self.n_from_right = int(cr[1] + cr[2])
        self.n_from_left = int(cl[2] + cl[3])
        s.N = N
        s.n_ghost = self.n_from_left + self.n_from_right
        # positions live in a buffer that only grows: no allocation per rebuild, and the pointer the force path
        # and the halo see stays put
        need = N + s.n_ghost
        if getattr(self, "_pos_buf", None) is None or self._pos_buf.shape[0] < need or self._pos_buf.dtype != new_pos.dtype:
            self._pos_buf = torch.zeros((int(need * 1.2) + 64, 4), dtype=new_pos.dtype, device=new_pos.device)
        self._pos_buf[:N] = new_pos
        s.pos = self._pos_buf[:need]
        s.vel = new_vel.contiguous()
        if s.force.shape[0] != N:
            s.force = torch.zeros((N, 4), dtype=s.dtype, device=s.pos.device)
            s.virial = torch.zeros(6 * N, dtype=s.dtype, device=s.pos.device)
        if self.transport_request != "torch" and self._native is None:
            self._try_native()
        self.exchange()

    def _classify_sort_device(self, N):
        """-> (counts of the 16 (destination, class) keys [int64, device], the stable order by key [int32, device])."""
        import ctypes as C
        from ._lib import lib, check
        s = self.sys
        dev = s.pos.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        st = getattr(self, "_plan_buf", None)
        if st is None or st["key"].shape[0] < N or st["bnd"].dtype != s.pos.dtype:
            st = self._plan_buf = {
                "key": torch.empty(int(N * 1.2) + 64, dtype=torch.int32, device=dev),
                "order": torch.empty(int(N * 1.2) + 64, dtype=torch.int32, device=dev),
                "scratch": torch.zeros(16 * ((int(N * 1.2) + 64 + 4095) // 4096), dtype=torch.int32, device=dev),
                "start": torch.empty(17, dtype=torch.int32, device=dev),
                "bnd": torch.as_tensor(self.bounds, dtype=s.pos.dtype, device=dev)}
        check(lib.htfs_slab_classify(s.pos.data_ptr(), s.scalar_code, N, st["bnd"].data_ptr(), self.world, self.rank,
                                     self.r_ghost, st["key"].data_ptr(), stream))
        check(lib.htfs_key_sort16(st["key"].data_ptr(), N, st["scratch"].data_ptr(), st["start"].data_ptr(),
                                  st["order"].data_ptr(), stream))
        cnt = (st["start"][1:] - st["start"][:-1]).to(torch.int64)
        return cnt, st["order"][:N]

    def _merge_segments_device(self, pv, stay, in_r, in_l):
        """[stay by class | from the right by class | from the left by class] -> one array by class (inside a class: stayed,
        from the right, from the left -- the order a stable sort of the class vector gives)."""
        import ctypes as C
        from ._lib import lib, check
        segs = [np.asarray(v, dtype=np.int64) for v in (stay, in_r, in_l)]
        src0 = np.concatenate([[0], np.cumsum([int(v.sum()) for v in segs])])
        src, dst, cnt = [], [], []
        off = 0
        for c in range(4):
            for k, v in enumerate(segs):
                src.append(int(src0[k] + v[:c].sum()))
                dst.append(off)
                cnt.append(int(v[c]))
                off += int(v[c])
        out = torch.empty_like(pv)
        U = C.c_uint * 12
        check(lib.htfs_segment_copy(out.data_ptr(), pv.data_ptr(), int(pv.shape[1]) * pv.element_size(), 12, U(*src), U(*dst), U(*cnt),
                                    C.c_void_p(torch.cuda.current_stream(pv.device).cuda_stream)))
        return out

    def _try_native(self):
        """Bring up the native transport and check it ONCE against the torch one (same slices, same ghosts)."""
        want, self.transport_request = self.transport_request, "torch"  # decided once
        s = self.sys
        try:
            from ._lib import lib
            if not s.pos.is_cuda or dist.get_backend(self.group) != "nccl" or not lib.htf_halo_available():
                if want == "native":
                    raise RuntimeError("native halo needs device positions, the nccl backend and a loadable librccl")
                return
            native, err = None, ""
            try:
                native = _NativeHalo(self.rank, self.world, self.group)
            except Exception as e:  # noqa: BLE001 -- every rank must reach the agreement below
                err = str(e)
            agreed = torch.tensor([0.0 if native is None else 1.0], device=s.pos.device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=self.group)
            if float(agreed.item()) != 1.0:
                raise RuntimeError("communicator bring-up failed on some rank" + (": " + err if err else ""))
            self.exchange()  # torch transport: the reference result
            ghosts = s.pos[s.N:s.N + s.n_ghost].clone()
            s.pos[s.N:s.N + s.n_ghost] = float("nan")
            self._native, self.transport = native, "native"
            # A failure of the native exchange on ONE rank must not let that rank skip the agreement below while
            # its peers wait in it (mismatched collectives = a hang): catch locally, fold into the flag, and have
            # every rank reach the all_reduce.
            same, err = False, ""
            try:
                self.exchange()
                same = torch.equal(s.pos[s.N:s.N + s.n_ghost], ghosts)
            except Exception as e:  # noqa: BLE001
                err = str(e)
                self._works = None
            ok = torch.tensor([1.0 if same else 0.0], device=s.pos.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if float(ok.item()) != 1.0:
                # (every rank takes this branch together; the torch transport's ghosts are restored)
                s.pos[s.N:s.N + s.n_ghost] = ghosts
                raise RuntimeError("native halo " + ("failed: " + err if err else "delivered different ghosts than torch.distributed (here or on a peer)"))
        except Exception as e:  # noqa: BLE001
            self._native, self.transport = None, "torch"
            if want == "native":
                raise
            self.transport_note = "native halo not used: %s" % (e,)

    def row_classes(self):
        """Class id (0..3, see rebuild) of every local row: a particle sorter may only permute
        rows inside a class, or the send slices and the interior range no longer mean anything."""
        c = torch.as_tensor(self.class_counts, dtype=torch.int64, device=self.sys.pos.device)
        return torch.repeat_interleave(torch.arange(4, device=c.device), c)

    def exchange_begin(self):
        """Post the per-step forward halo (ghost positions from their owners).  The messages
        are slices of ``pos`` itself; on RCCL the transfer runs on the communicator's stream
        after everything already queued on the current stream, so interior rows can be
        evaluated meanwhile.  Nothing may write ``pos`` until exchange_end()."""
        if self.world == 1 or self.send_left is None:
            return
        s = self.sys
        N = s.N
        to_left = s.pos[self.send_left[0]:self.send_left[1]]
        to_right = s.pos[self.send_right[0]:self.send_right[1]]
        from_left = s.pos[N:N + self.n_from_left]                 # R> of my left neighbor
        from_right = s.pos[N + self.n_from_left:N + s.n_ghost]    # L> of my right neighbor
        if self._native is not None:
            self._native.begin(s.pos, self.left, self.right, self.send_left, self.send_right,
                               (N, N + self.n_from_left), (N + self.n_from_left, N + s.n_ghost))
            self._works = [self._native]
            return
        self._works = self._post(to_left, to_right, from_right, from_left)

    def exchange_end(self):
        """Make the current stream wait for the halo posted by exchange_begin()."""
        if self._works is not None:
            for w in self._works:
                w.wait()
            self._works = None

    def exchange(self):
        """Per-step forward halo, blocking form."""
        self.exchange_begin()
        self.exchange_end()
