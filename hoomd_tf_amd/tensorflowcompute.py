"""``tfcompute``: the counterpart of ``hoomd/htf/tensorflowcompute.py`` -- wires a
``SimModel`` into the MD loop as a force.

Where the reference constructs ``_htf.TensorflowCompute[GPU]`` and is called back from C++
once per batch (``_finish_update``), this class owns an ``htf_ctx`` (the C-ABI
TensorflowCompute) and has two execution paths:

* traced (default for declarative models): the first step runs ``model.compute`` eagerly;
  if it consisted of a single ``compute_nlist_forces`` over the full neighbor tensor and
  nothing else needs saving, the lowered potential is installed in the context and every
  later step is ONE C call (``htf_compute_forces``): no Python per batch, like a traced
  ``tf.function``.
* eager: ``_finish_update(batch_index)`` per batch exactly as the reference orders it:
  compute_inputs -> model(inputs) -> outputs capture -> compute_outputs.
"""
import os

import numpy as np
import torch

from . import _lib, ops, simmodel, standin
from .simmodel import SimModel


class tfcompute:
    """tensorflowcompute.py:20-36."""

    def __init__(self, model):
        self.model = model
        self.cpp_force = None
        self._nlist = None
        self.map_types = set()
        # traced path (htf_config.fused): 2 = one kernel writes the pair-vector tensor and evaluates
        # it while it is in registers (default); 0 = build kernel, then evaluator kernel; 1 = opt-in,
        # no tensor at all (get_nlist_array() then has no side buffer to read)
        self.fused = 2

    def attach(self, nlist=None, r_cut=0, period=1, batch_size=None, train=False, save_output_period=None):
        """tensorflowcompute.py:38-188 (same arguments and error behaviour)."""
        sim = standin.current_simulation()
        if sim is None:
            raise RuntimeError('Must initialize hoomd first')
        self.sim = sim
        self.system = sim.system
        self.enabled = True
        self.force_name = 'tfcompute'
        self.r_cut = float(r_cut)
        self.batch_size = 0 if batch_size is None else int(batch_size)
        self.period = int(period)
        self.save_output_period = save_output_period
        self.outputs = None
        self._calls = 0
        self._output_offset = 0
        if self.model.output_forces:
            self._output_offset = 1
        if self.model.virial:
            self._output_offset = 2
        if train:
            # figure out from losses (tensorflowcompute.py:87-95)
            if getattr(self.model, 'loss', None) is None:
                raise ValueError('SimModel has not been compiled')
            for i, l in enumerate(self.model.loss):
                if l is None:
                    break
            else:
                i = len(self.model.loss)
            self._output_offset = i
        self.train = train
        self._ref_forces = []
        self._labels = None
        self._opt_state = None
        if isinstance(self.model, simmodel.MolSimModel):  # tensorflowcompute.py:103-111
            if self.batch_size != 0:
                raise ValueError('Cannot batch by molecule and by batch_number')
            if nlist is not None and getattr(nlist, "domain", None) is not None and nlist.domain.world > 1:
                raise ValueError('Molecular batches are not supported with spatial decomposition (MPI)')
            self._disable_sorter(nlist)
        self.nneighbor_cutoff = self.model.nneighbor_cutoff
        if nlist is not None:
            nlist.subscribe(self.rcut)
            self._nlist = nlist
            if self.model._map_nlist:  # tensorflowcompute.py:170-174
                nlist.type_split = self._map_typeid_start
                self._disable_sorter(nlist)
        elif self.nneighbor_cutoff != 0:
            raise ValueError('Must provide an nlist if you have nneighbor_cutoff > 0')
        self.force_mode_code = _lib.HTF_TF2HOOMD if self.model.output_forces else _lib.HTF_HOOMD2TF
        s = self.system
        self.dtype = s.dtype  # isDoublePrecision() (tensorflowcompute.py:166-168)
        self.cpp_force = ops.Context(r_cut=self.r_cut if self.nneighbor_cutoff else 0.0,
                                     nneighs=self.nneighbor_cutoff, period=self.period,
                                     batch_size=self.batch_size, scalar_dtype=s.dtype,
                                     check_nlist=self.model.check_nlist, virial=self.model.virial, max_n=s.N,
                                     fused=self.fused)
        # ForceCompute::m_force / m_virial of this compute
        self.force = torch.zeros((s.N, 4), dtype=s.dtype, device=s.device)
        self.virial = torch.zeros(6 * s.N, dtype=s.dtype, device=s.device)
        self._plan = None
        self._plan_folded = ()   # (weight, _version) pairs whose values are constants of the plan's generated kernel
        self._bplan = None  # EDS-biased model replayed as one kernel (see _maybe_install_plan)
        self._tplan = None  # a training step replayed without calling compute() (see _maybe_install_train_plan)
        self._train_seen = None
        self._post_ops = []  # replayable observables of the planned step (compute_rdf -> MeanTensor), run after the force kernel
        self._post_src = None
        self._ctx_ran = False
        self.model._plan = None
        self.log_name = 'tensorflow'  # m_log_name, TensorflowCompute.cc:62
        if self.force_mode_code == _lib.HTF_TF2HOOMD:
            sim.forces.append(self)   # hoomd.context.current.forces.append(self)
        else:
            # tensorflowcompute.py:183-188: hoomd2tf computes run from the integrator's half-step
            # hook, i.e. once the step's forces exist; the stand-in calls sim.computes right there
            if sim.integrator is None:
                raise ValueError('Must have integrator set to receive forces')
            sim.computes.append(self)

    def rcut(self):
        """tensorflowcompute.py:284-305: the cutoff this compute subscribes to the nlist (the
        all-atom / mapped type-pair exclusion is the nlist's ``type_split``)."""
        return self.r_cut

    @staticmethod
    def _disable_sorter(nlist):
        """tensorflowcompute.py:190-196: molecular batching / mapped beads need a fixed order."""
        if nlist is not None:
            nlist.sort_particles = False

    def enable_mapped_nlist(self, system, mapping_fxn):
        """tensorflowcompute.py:198-263: put the M coarse-grained beads of ``mapping_fxn(positions
        [N,4], [Lx,Ly,Lz]) -> [M,4]`` (torch; column 3 = bead type) into the simulation behind
        the N all-atom particles, with type ids offset so the neighbor search never pairs the
        two kinds.  Returns (aa_group, mapped_group); integrate the first only."""
        AAN = system.N
        dtype = self.model.dtype
        pos = ops.copy_positions(system.pos, offset=0, N=AAN, unstuff4=True).to(dtype)
        L = system.box3x3[1] - system.box3x3[0]
        cg = mapping_fxn(pos, torch.as_tensor(L, dtype=dtype, device=pos.device))
        cg = cg.detach().cpu().numpy()
        M = cg.shape[0]
        typeid = system.types_numpy()
        map_typeid_start = int(np.max(typeid[M:])) + 1  # (sic) tensorflowcompute.py:234
        new_types = cg[:, 3].astype(np.int32) + map_typeid_start
        for i in new_types:
            self.map_types.add(int(i))
        system.append_particles(cg[:, :3], new_types)
        self.model._map_nlist = True
        self.model._map_fxn = mapping_fxn
        self.model._map_i = AAN
        self._map_typeid_start = map_typeid_start
        if self._nlist is not None:
            self._nlist.type_split = map_typeid_start
            self._nlist._ref = None  # update_rcut(): rebuild with the new types
        return standin.Group(0, AAN), standin.Group(AAN, M)

    def get_log_value(self, quantity, timestep):
        """TensorflowCompute::getLogValue (.cc:376-395): 'tensorflow' -> compute(timestep), then
        the energy sum over the particles (all ranks when the box is decomposed)."""
        if quantity != self.log_name:
            raise RuntimeError('Error getting log value: tensorflow:%s is not a valid log quantity' % quantity)
        self.compute(timestep)
        e = ops.energy_sum(self.force)
        domain = getattr(self._nlist, "domain", None)
        if domain is not None and domain.world > 1:
            import torch.distributed as dist
            dist.all_reduce(e, group=domain.group)
        return float(e.item())

    def update_coeffs(self):
        pass

    def disable(self):
        """hoomd compute.disable(): take this compute out of the step loop."""
        for lst in (self.sim.forces, self.sim.computes):
            if self in lst:
                lst.remove(self)
        self.enabled = False

    def enable(self):
        if not self.enabled:
            (self.sim.forces if self.force_mode_code == _lib.HTF_TF2HOOMD else self.sim.computes).append(self)
            self.enabled = True

    def set_reference_forces(self, *forces):
        """tensorflowcompute.py:265-282."""
        if self.force_mode_code == _lib.HTF_TF2HOOMD:
            raise ValueError('Only valid to set reference forces if mode is hoomd2tf')
        for f in forces:
            if not hasattr(f, 'force') and not hasattr(f, 'cpp_force'):
                raise ValueError('given force does not seem like a hoomd force')
            self._ref_forces.append(f)

    # ------------------------------------------------------------------ per step
    def _arrays(self):
        s, nl = self.system, self._nlist
        if nl is not None:
            return self.cpp_force.make_arrays(s.pos, s.N, nl.n_neigh, nl.head_list, nl.nlist, s.box, self.force,
                                              self.virial, s.N)
        return self.cpp_force.make_arrays(s.pos, s.N, None, None, None, s.box, self.force, self.virial, s.N)

    def graph_safe(self):
        """True when a step of this compute is a fixed sequence of launches (the installed one-kernel plan, every
        step, no training, no output capture, no host-side precompute): Simulation.run may replay it from a hipGraph."""
        # (check_nlist reads a count back inside htf_compute_forces -- a host copy and a synchronize, illegal under stream
        #  capture -- so a model that asks for it steps eagerly)
        return (self._plan is not None and self.model._plan is self._plan and self.period == 1 and not self.train
                and not self._plan_is_stale()
                and not getattr(self.model, "check_nlist", False)
                and not self.model._map_nlist and getattr(self._nlist, "domain", None) is None
                and self.force.shape[0] == self.system.N and not getattr(self, "save_output_period", None))

    def _plan_is_stale(self):
        """A weight whose value was folded into the plan's generated kernel (simmodel.PairExpr._with) has been written since."""
        return any(t._version != v for t, v in getattr(self, "_plan_folded", ()))

    def graph_key(self):
        # the potential object AND its handle / parameter version (a refresh re-images device weights in place: same handle,
        # same addresses -- kernels resolve them at launch; a new Potential is a new handle), the context, the output arrays
        p = self._plan
        return (id(p), getattr(getattr(p, "handle", None), "value", None), getattr(p, "version", 0), id(self.cpp_force),
                self.force.data_ptr(), self.virial.data_ptr(), int(self.fused), int(self.nneighbor_cutoff), float(self.r_cut))

    def compute(self, timestep):
        """ForceCompute::compute -> TensorflowCompute::computeForces (.cc:129-216)."""
        if timestep % self.period != 0:
            return
        if self.model._map_nlist:
            self.model.precompute(self.system)  # startUpdate -> _start_update, .cc:228-241
        if self._nlist is not None:
            self._nlist.compute(timestep)  # m_nlist->compute(timestep), .cc:162-163
        domain = getattr(self._nlist, "domain", None)
        if self.force.shape[0] != self.system.N:
            # MaxParticleNumberChange -> reallocate (.cc:88): particles migrated between ranks
            s = self.system
            self.force = torch.zeros((s.N, 4), dtype=s.dtype, device=s.device)
            self.virial = torch.zeros(6 * s.N, dtype=s.dtype, device=s.device)
        if self._plan is not None and self._plan_is_stale():
            self._plan = self.model._plan = None      # (re-traced below with the weights' present values)
            self._plan_folded = ()
        if self._plan is not None and getattr(self, "_plan_weights", None) is not None:
            self._plan_weights.refresh_if_stale()     # (weights that are kernel ARGUMENTS: a 4-byte copy each, the plan stays)
        if self._plan is not None and self.model._plan is self._plan:
            self._calls += 1
            # interior rows while the ghost halo is in flight, boundary rows after it
            self.cpp_force.compute_forces_overlapped(timestep, self._arrays(), domain)
            self._ctx_ran = True
            if self._post_ops:
                src = self._post_src
                if src is None or src[0].shape[0] != self.system.N:
                    src = self._post_src = (self.cpp_force.nlist_buffer(self.system.N, self.system.device),
                                            self.cpp_force.positions_buffer(self.system.N, self.system.device))
                for op in self._post_ops:
                    op(src[0], src[1])
            return
        if domain is not None:
            domain.exchange_end()
        if self._bplan is not None and self.model._plan is self._bplan:
            self._calls += 1
            self._run_biased_plan()
            return
        if self._tplan is not None:
            self._calls += 1
            self._run_train_plan()
            return
        s = self.system
        bs = s.N if self.batch_size == 0 else self.batch_size
        simmodel._trace_log().clear()
        if self.train:
            self._stage_labels()
        nbatch = 0
        for i in range(s.N // bs + 1):
            offset = i * bs
            n = min(s.N - offset, bs)
            if n < 1:
                break
            self._finish_update(i, offset, n)
            nbatch += 1
        if not self.train:
            self._maybe_install_plan(nbatch)
        else:
            self._maybe_install_train_plan(nbatch)
        simmodel._trace_log().clear()

    def _stage_labels(self):
        """TensorflowCompute.cc:177-187: labels = HOOMD net force, or the sum of the selected
        reference forces (sumReferenceForces, :250-269), staged once per step for all batches."""
        s = self.system
        if not self._ref_forces:
            self._labels = self.sim.net_force
        else:
            lab = self._ref_forces[0].force.clone()
            for f in self._ref_forces[1:]:
                ops.add_scalar4(lab, f.force)
            self._labels = lab

    def _train_generic(self, output, offset, n):
        """train_on_batch for a generic (torch-op) model: MSE over the loss-list outputs by
        autograd -- the forces were built with create_graph, so the gradient flows through them
        into the model's torch parameters -- and the compiled optimizer's torch twin."""
        m = self.model
        params = m.parameters()
        pred = output[0]
        d = pred.to(torch.float32) - self._labels[offset:offset + n].to(torch.float32)
        loss = (d * d).mean()
        if self._opt_state is None:
            self._opt_state = torch.zeros(_lib.OPT_STATE_FLOATS, dtype=torch.float32, device=self.system.device)
            if m.metrics:
                m.metrics[0].state = self._opt_state
        if params:
            if getattr(self, "_torch_opt", None) is None:
                self._torch_opt = m.optimizer.torch(params)
            self._torch_opt.zero_grad(set_to_none=True)
            loss.backward()
            self._torch_opt.step()
        self._opt_state[18] += loss.detach()
        self._opt_state[19] += 1.0
        self._opt_state[20] = loss.detach()

    def _train_on_batch(self, nlist_t, offset, n, fused_entries):
        """model.train_on_batch(x=inputs, y=labels) (tensorflowcompute.py:366-370) for a model
        whose forces come from one trainable closed-form layer: loss + parameter gradient in
        one sweep (htf_train_pair_grad), optimizer step on the device (htf_optimizer_step)."""
        entries = [e for e in fused_entries if e.get("layer") is not None]
        m = self.model
        n_theta = sum(int(e["layer"].make_trainable(self.system.device).numel()) for e in entries[:1])
        if self._opt_state is None:
            self._opt_state = torch.zeros(ops.optimizer_state_floats(n_theta), dtype=torch.float32,
                                          device=self.system.device)
            if m.metrics:
                m.metrics[0].state = self._opt_state
        if not entries:
            # no trainable weights (test_force_output trains LJModel): train_on_batch only
            # evaluates the loss; O(N) elementwise on the [n, 4] prediction
            if not fused_entries:
                raise ValueError('training needs the model forces to come from compute_nlist_forces')
            d = fused_entries[-1]["forces"].to(torch.float32) - self._labels[offset:offset + n].to(torch.float32)
            self._opt_state[18] += (d * d).mean()
            self._opt_state[19] += 1.0
            return
        if len(entries) != 1:
            raise ValueError('training needs exactly one compute_nlist_forces over a trainable layer '
                             '(LJLayer, WCARepulsion, PairMLP); found %d' % len(entries))
        layer = entries[0]["layer"]
        theta = layer.make_trainable(self.system.device)
        pot = layer.potential()
        if getattr(self, "_opt_desc", None) is None:
            self._opt_desc = m.optimizer.desc(layer.nonneg_mask, layer.l1_reg)
        labels = self._labels[offset:offset + n]
        if nlist_t is None:     # (the training plan, from the index list: whole-system batches only)
            s_, nl_ = self.system, self._nlist
            accum = ops.train_pair_grad_list(pot, s_.pos, nl_.n_neigh, nl_.head_list, nl_.nlist, s_.box, self.r_cut,
                                             self.nneighbor_cutoff, labels, n_local=s_.N)
        else:
            accum = ops.train_pair_grad(pot, nlist_t, labels)
        n_total = float(n)
        domain = getattr(self._nlist, "domain", None)
        if domain is not None and domain.world > 1:
            # data-parallel training over the slabs: ONE all-reduce of [loss, gradient, count]
            # per batch (26 KB for the pair-MLP) keeps the replicas' weights identical; the
            # reference trains each MPI rank's copy independently
            import torch.distributed as dist
            dist.all_reduce(accum, group=domain.group)
            # the global batch: every rank's share of this batch, from the per-rank particle counts the last rebuild left on
            # the host (no count in the message, no read-back -- the training step enqueues and returns)
            bs = self.batch_size
            if not bs and getattr(domain, "n_global", None):
                n_total = float(domain.n_global)      # (fixed-capacity bricks: particles are conserved, no count to gather)
            else:
                n_total = float(sum(min(max(nq - offset, 0), bs) if bs else nq for nq in domain.local_counts))
        elif domain is not None and getattr(domain, "fixed_capacity", False) and not self.batch_size:
            n_total = float(domain.n_global) / float(np.prod(domain.grid))   # (a replica brick: rows are a capacity, particles are not)
        ops.optimizer_step(theta, accum, 1.0 / (4.0 * n_total), self._opt_state, self._opt_desc)
        if hasattr(layer, "after_update"):
            layer.after_update()  # pair-MLP: operand images <- theta, on the device
        self._train_potential = pot  # keep alive until the stream has consumed it

    def _maybe_install_train_plan(self, nbatch):
        """A training step whose model is ONE trainable pair energy on the step's own neighbor tensor (LJLayer, WCARepulsion, PairMLP,
        a traced energy with weights) is, from the second step on, a fixed sequence -- pair vectors, training sweep (prediction, loss,
        parameter gradient), optimizer kernel -- and is replayed without calling compute(): no trace, no separate forward sweep
        (the training sweep predicts by itself), no per-step host -> device copy of the box (a blocking copy: the host could not
        run ahead of the device).  Same rule as the inference plan: model code does not run again once the plan is installed, so
        anything else in the step -- an observable, saved outputs, a mapped or masked list, batches -- keeps the eager path.
        HTF_NO_TRAIN_PLAN=1 keeps every step eager."""
        seen, self._train_seen = self._train_seen, None
        if seen is None or nbatch != 1 or os.environ.get("HTF_NO_TRAIN_PLAN") == "1":
            return
        entries, nlist_t, output = seen
        if (len(entries) == 1 and entries[0].get("layer") is not None and "potential" in entries[0] and not entries[0]["virial"]
                and entries[0]["nlist"].tensor is nlist_t and len(output) > 0 and output[0] is entries[0]["forces"]
                and not self.save_output_period and not self.model._map_nlist and self.nneighbor_cutoff > 0 and not self.model.check_nlist
                and not isinstance(self.model, simmodel.MolSimModel)):
            pot = entries[0]["potential"]
            # closed forms and traced energies train straight from the index list (htf_train_pair_grad_list): the step neither
            # writes nor re-reads the [N, NN, 4] tensor; the pair-MLP's sweep reads the tensor.  HTF_TRAIN_FROM_TENSOR=1: always.
            from_list = (pot.kind in (_lib.POT_LJ_PARAM, _lib.POT_WCA, _lib.POT_RINV_POLY, _lib.POT_JIT)
                         and os.environ.get("HTF_TRAIN_FROM_TENSOR") != "1")
            self._tplan = {"layer": entries[0]["layer"], "potential": pot, "from_list": from_list}

    def _run_train_plan(self):
        s, nl = self.system, self._nlist
        self._stage_labels()
        if self._tplan["from_list"]:
            self._last = None          # (get_nlist_array / get_positions_array build the side buffers when somebody asks)
            self._train_on_batch(None, 0, s.N, [self._tplan])
            return
        pv = ops.build_pair_vectors(s.pos, nl.n_neigh, nl.head_list, nl.nlist, s.box, self.r_cut, self.nneighbor_cutoff,
                                    offset=0, batch_size=s.N, n_local=s.N, out_dtype=torch.float32)
        self._last = (pv, ops.copy_positions(s.pos, offset=0, N=s.N, unstuff4=True), 0, s.N)
        self._train_on_batch(pv, 0, s.N, [self._tplan])

    def _side_buffers(self):
        """(nlist tensor, positions, offset, n) of the last batch; after a training step that ran from the index list they are
        built here, from the arrays as they stand."""
        if self._last is None:
            s, nl = self.system, self._nlist
            pv = ops.build_pair_vectors(s.pos, nl.n_neigh, nl.head_list, nl.nlist, s.box, self.r_cut, self.nneighbor_cutoff,
                                        offset=0, batch_size=s.N, n_local=s.N, out_dtype=torch.float32)
            self._last = (pv, ops.copy_positions(s.pos, offset=0, N=s.N, unstuff4=True), 0, s.N)
        return self._last

    def _maybe_install_plan(self, nbatch):
        log = simmodel._trace_log()
        fused = [e for e in log if "potential" in e]
        # observables that can redo themselves on the device (compute_rdf of the step's tensor feeding a MeanTensor, example
        # 01): the step is still ONE fixed launch sequence -- force kernel, then their replays on the tensor it wrote
        obs = [e for e in log if e.get("observable")]
        # Only what a traced consumer reads is worth replaying: a bare compute_rdf whose result nothing traced takes -- a list
        # append, a torch in-place accumulation in model code -- keeps the model on the eager path, as before: after
        # the plan is installed SimModel.compute is never called again, and only htf.MeanTensor updates survive it.
        # Under domain decomposition the replays run on the rank's own rows, as the eager path (and the reference: every MPI rank
        # runs its own model on its local particles, test_mpi_tensorflow.py) does: the global RDF is the sum over ranks.
        post = []
        consumed = [v for e in obs if e.get("op") == "metric_update" and e.get("replay") is not None for v in e.get("inputs", ())]
        fed = all(any(o is v for o in e.get("outputs", ()) for v in consumed) for e in obs if e.get("op") == "compute_rdf")
        if (obs and fed and nbatch == 1 and len(log) == 1 + len(obs) and all(e.get("replay") is not None for e in obs)
                and int(self.fused) == 2 and not self.model._map_nlist):
            post = [e["replay"] for e in obs]
        plain = [e for e in log if not e.get("observable")] if post else log
        self._post_ops, self._post_src = [], None
        if (self.force_mode_code == _lib.HTF_TF2HOOMD and len(plain) == nbatch and len(fused) == nbatch
                and all(e.get("is_output") for e in fused) and not self.save_output_period
                and all(e["virial"] == bool(self.model.virial) for e in fused)
                and len({id(e["potential"]) for e in fused}) == 1
                and (not fused[0].get("extra_potentials") or (int(self.fused) == 2 and nbatch == 1 and not self.model._map_nlist))):
            self._plan = fused[0]["potential"]
            self._plan_folded = tuple(fused[0].get("folded", ()))
            lay = fused[0].get("layer")
            self._plan_weights = lay if hasattr(lay, "refresh_if_stale") else None
            self.model._plan = self._plan
            self.cpp_force.set_potential(self._plan)
            # an energy of several row terms (simmodel._row_forces): the further terms from the tensor the step's kernel wrote
            self._post_ops = post + [self._accumulate_term(p_, typed) for p_, typed in (fused[0].get("extra_potentials") or ())]
        else:
            # config C4's shape: closed-form base energy + alpha * soft-RDF CV (+ observables nobody
            # saves).  Replayed as htf_build_eval_forces2: tensor, both force sets and the CV partials
            # from one kernel, then the device-side EDS update and the force assembly.
            biased = [e for e in log if "biased" in e]
            rest = [e for e in log if "biased" not in e and not e.get("observable")]
            if (self.force_mode_code == _lib.HTF_TF2HOOMD and nbatch == 1 and len(biased) == 1 and not rest
                    and biased[0].get("is_output") and not self.save_output_period and not self.model.virial
                    and not self.model.check_nlist and int(self.fused) == 2
                    and getattr(self._nlist, "domain", None) is None and not self.model._map_nlist):
                self._bplan = dict(biased[0]["biased"])
                self.model._plan = self._bplan
        log.clear()

    def _accumulate_term(self, pot, typed):
        """A replayable post-op of the planned step: force += the streaming evaluator of one more term on the step's own tensor."""
        state = {"buf": None}

        def op(nl_buf, pos_buf):
            n = nl_buf.shape[0]
            if state["buf"] is None or state["buf"].shape[0] != n:
                state["buf"] = torch.empty((n, 4), dtype=self.force.dtype, device=self.force.device)
            ops.eval_forces(pot, nl_buf, out=state["buf"], positions=pos_buf if typed else None)
            ops.add_scalar4(self.force, state["buf"])
        op.potential = pot   # (kept alive with the plan)
        return op

    def _run_biased_plan(self):
        s, nl, bp = self.system, self._nlist, self._bplan
        NN = self.nneighbor_cutoff
        if bp.get("n") != s.N:
            bp["n"] = s.N
            bp["pv"] = torch.empty((s.N, NN, 4), dtype=torch.float32, device=s.device)
            bp["fb"] = torch.empty((s.N, 4), dtype=s.dtype, device=s.device)
            bp["np"] = ops.num_partials_fused(s.N)
            bp["partials"] = torch.empty(bp["np"], dtype=torch.float32, device=s.device)
            bp["cv_value"] = torch.empty(1, dtype=torch.float32, device=s.device)
        ops.build_eval_forces2(bp["pot_a"], bp["pot_b"], s.pos, nl.n_neigh, nl.head_list, nl.nlist, s.box, self.r_cut, NN,
                               n_local=s.N, partials=bp["partials"], pair_vectors=bp["pv"], out_a=self.force, out_b=bp["fb"])
        ops.reduce_partials(bp["partials"], bp["np"], 1.0 / s.N, bp["cv_value"])
        bp["cv"].value = bp["cv_value"]
        eds = bp["eds"]  # EDSLayer.__call__ without the trace entry
        _lib.check(_lib.lib.htf_eds_update(eds.state.data_ptr(), bp["cv_value"].data_ptr(), eds.set_point, eds.period,
                                           eds.learning_rate, eds.cv_scale, ops._stream(eds.state)))
        ops.bias_combine(self.force, bp["fb"], eds.state[2:3], bp["cv_value"])
        self._last = (bp["pv"], ops.copy_positions(s.pos, offset=0, N=s.N, unstuff4=True), 0, s.N)

    def _finish_update(self, batch_index, offset, n):
        """tensorflowcompute.py:313-345 (inference branch)."""
        if batch_index == 0:
            self._calls += 1
        s, nl, m = self.system, self._nlist, self.model
        NN = self.nneighbor_cutoff
        if NN > 0:
            nlist_t = ops.build_pair_vectors(s.pos, nl.n_neigh, nl.head_list, nl.nlist, s.box, self.r_cut, NN,
                                             offset=offset, batch_size=n, n_local=s.N, out_dtype=torch.float32)
        else:
            nlist_t = torch.zeros((1, 1, 4), dtype=m.dtype, device=s.device)  # simmodel.py:179
        pos_t = ops.copy_positions(s.pos, offset=offset, N=n, unstuff4=True)
        box_t = torch.as_tensor(s.box3x3, dtype=s.dtype, device=s.device)
        self._last = (nlist_t, pos_t, offset, n)
        inputs = m.compute_inputs(nlist_t, pos_t, box_t)
        mark = len(simmodel._trace_log())
        simmodel._trace.training_graph = bool(self.train)  # generic route: forces stay differentiable
        try:
            output = m(inputs, self.train)
        finally:
            simmodel._trace.training_graph = False
        fused_entries = [e for e in simmodel._trace_log()[mark:] if "potential" in e]
        for e in simmodel._trace_log()[mark:]:
            if "forces" in e and len(output) > 0 and output[0] is e["forces"] and e["nlist"] is inputs[0]:
                e["is_output"] = True
            if e.get("op") == "compute_rdf":
                tt = e.get("type_tensor")
                if e.get("nlist") is not inputs[0]:
                    e["replay"] = None  # an RDF of something else than the step's neighbor tensor (a masked or mapped list): eager
                elif tt is not None and not (isinstance(tt, torch.Tensor) and isinstance(inputs[1], torch.Tensor)
                                             and tt.dim() == 1 and tt.shape[0] == n and tt.stride(0) == 4
                                             and tt.data_ptr() == inputs[1].data_ptr() + 3 * inputs[1].element_size()
                                             and inputs[1].is_contiguous()):
                    e["replay"] = None  # types from anywhere but positions[:, 3] of this step's positions tensor: eager
        if self.save_output_period and self._calls % self.save_output_period == 0:
            extra = [_np(o)[np.newaxis, ...] for o in output[self._output_offset:]]
            if self.outputs is None:
                self.outputs = extra
            else:
                self.outputs = [np.append(o1, o2, axis=0) for o1, o2 in zip(self.outputs, extra)]
        if self.train:
            if any(e.get("op") == "generic" for e in simmodel._trace_log()[mark:]):
                self._train_generic(output, offset, n)
            else:
                self._train_on_batch(nlist_t, offset, n, fused_entries)
                self._train_seen = (list(simmodel._trace_log()[mark:]), nlist_t, output)
            return
        if self.force_mode_code == _lib.HTF_TF2HOOMD:
            f = SimModel.compute_outputs(_t(output[0]).detach(), s.dtype)
            self.force[offset:offset + n] = f
            if m.virial:
                v = _t(output[1]).to(s.dtype).reshape(n, 9).contiguous()
                ops.add_virial(self.virial[offset:], v, n, s.N)  # receiveVirial, .cc:200-204
                self._last_virial = (offset, v)

    # ------------------------------------------------------------------ array getters
    def get_positions_array(self):
        """tensorflowcompute.py:372-375 (last batch's positions side buffer)."""
        if self._ctx_ran:
            n = self.system.N if self.batch_size == 0 else min(self.batch_size, self.system.N)
            return self.cpp_force.positions_buffer(n, self.system.device).double().cpu().numpy()
        return self._side_buffers()[1].double().cpu().numpy()

    def get_nlist_array(self):
        """tensorflowcompute.py:377-381 -> [B, NN, 4]."""
        if self._ctx_ran and int(self.fused) == 1:
            raise RuntimeError('fused mode keeps the pair vectors in registers; there is no nlist buffer to read')
        if self._ctx_ran:
            n = self.system.N if self.batch_size == 0 else min(self.batch_size, self.system.N)
            return self.cpp_force.nlist_buffer(n, self.system.device).double().cpu().numpy()
        return self._side_buffers()[0].double().cpu().numpy().reshape(-1, self.nneighbor_cutoff, 4)

    def get_forces_array(self):
        """tensorflowcompute.py:383-386."""
        return self.force.double().cpu().numpy()

    def get_virial_array(self):
        """tensorflowcompute.py:388-392: the [B, 9] side buffer."""
        if self._ctx_ran:
            n = self.system.N if self.batch_size == 0 else min(self.batch_size, self.system.N)
            return self.cpp_force.virial_buffer(n, self.system.device).double().cpu().numpy().reshape(-1, 9)
        if getattr(self, "_last_virial", None) is not None:
            return self._last_virial[1].double().cpu().numpy().reshape(-1, 9)
        return np.zeros((self.system.N, 9))


def _t(x):
    return x.tensor() if hasattr(x, "tensor") and callable(x.tensor) else x


def _np(x):
    x = _t(x)
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)
