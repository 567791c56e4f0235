"""`--workload generic-lj`: what leaving the lowered model zoo costs -- traced expressions as generated kernels against the same models
written in plain torch ops (the generic autograd route) and against lowered LJ."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, PROF_EVERY, ROOT, algorithmic_bytes, gpu_state, make_potential  # noqa: F401


def run_generic_lj(args, htf, standin, dev):
    """What leaving the zoo costs: the reference's defining capability is an ARBITRARY compute() (htf/simmodel.py:87-121) whose
    forces come from tf.gradients (simmodel.py:526-555).  Here a model outside the lowered closed forms / MLPs runs as torch
    eager ops on the zero-copy [N, NN, 4] tensor with torch.autograd for the forces.  The same LJModel twice, at
    C2 (32 768) and C3 (131 072) size, through tfcompute: written with the htf.* expression layer (lowered to the one-kernel step,
    replayed without Python) and written in plain torch ops (generic route: build kernel + ~20 eager ops + autograd every step)."""
    NN, rcut = args.nn, args.rcut

    class LJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            rinv = htf.nlist_rinv(nlist)
            inv_r6 = rinv**6
            p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
            energy = htf.reduce_sum(p_energy, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchLJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6  # nlist_rinv op for op (simmodel.py:618-635)
            r = torch.sqrt(torch.sum((nlist[:, :, :3] + delta / 3 / 10) ** 2, dim=2))
            rinv = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            inv_r6 = rinv ** 6
            p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
            energy = torch.sum(p_energy, dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    class MorseModel(htf.SimModel):
        """Outside the zoo, written with htf.* ops: a Morse well (D = 1, a = 5, r0 = 1.122) masked to the list's live slots.
        Traced into a generated kernel (HTF_POT_JIT, hoomd_tf_amd/codegen.py), replayed as the one-kernel step."""
        def compute(self, nlist, positions, box):
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            live = htf.cast(htf.nlist_rinv(nlist) > 0.0, torch.float32)
            x = 1.0 - htf.exp(-5.0 * (r - 1.122))
            energy = htf.reduce_sum(0.5 * live * (x * x - 1.0), axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class YukawaLJModel(htf.SimModel):
        """Outside the zoo: LJ plus a screened Coulomb term 0.5 exp(-r) / r (traced; generated kernel)."""
        def compute(self, nlist, positions, box):
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            s = htf.nlist_rinv(nlist)
            energy = htf.reduce_sum(2.0 * (s ** 12 - s ** 6) + 0.25 * htf.exp(-1.0 * r) * s, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchYukawaLJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            energy = torch.sum(2.0 * (s ** 12 - s ** 6) + 0.25 * torch.exp(-1.0 * r) * s, dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    KA_EPS, KA_SIG = [1.0, 1.5, 1.5, 0.5], [1.0, 0.8, 0.8, 0.88]   # Kob-Andersen 80:20 binary LJ: AA, AB, BA, BB

    class MixtureModel(htf.SimModel):
        """Outside the zoo AND typed: a binary LJ mixture whose epsilon and sigma are looked up by species pair -- tf.gather on
        ti * 2 + tj, the way a multi-component model is written against the reference -- traced into ONE generated kernel."""
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            idx = htf.cast(positions[:, 3], torch.int32)[:, None] * 2 + htf.cast(nlist[:, :, 3], torch.int32)
            q = (htf.gather(KA_SIG, idx) * s) ** 6
            energy = htf.reduce_sum(2.0 * htf.gather(KA_EPS, idx) * (q * q - q), axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchMixtureModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            idx = (positions[:, 3:4] * 2 + nlist[:, :, 3]).detach().long()
            q = (torch.tensor(KA_SIG, device=s.device)[idx] * s) ** 6
            energy = torch.sum(2.0 * torch.tensor(KA_EPS, device=s.device)[idx] * (q * q - q), dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    class IonicModel(htf.SimModel):
        """Outside the zoo, typed, with a special function: LJ cores plus the real-space part of Ewald / damped-shifted-force
        electrostatics between +1 / -1 species, q_i q_j erfc(alpha r) / r, the charges gathered by species pair."""
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            idx = htf.cast(positions[:, 3], torch.int32)[:, None] * 2 + htf.cast(nlist[:, :, 3], torch.int32)
            qq = htf.gather([1.0, -1.0, -1.0, 1.0], idx)
            energy = htf.reduce_sum(2.0 * (s ** 12 - s ** 6) + 0.5 * 2.0 * qq * htf.erfc(0.35 * r) * s, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchIonicModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            idx = (positions[:, 3:4] * 2 + nlist[:, :, 3]).detach().long()
            qq = torch.tensor([1.0, -1.0, -1.0, 1.0], device=s.device)[idx]
            energy = torch.sum(2.0 * (s ** 12 - s ** 6) + 0.5 * 2.0 * qq * torch.erfc(0.35 * r) * s, dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    def one(lattice, cells, model_cls, steps):
        pos, L, a = (standin.sc_positions if lattice == "sc" else standin.fcc_positions)(cells, 0.8442)
        rng = np.random.default_rng(7)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        types = (rng.random(len(pos)) < 0.2).astype(np.int32) if model_cls in (MixtureModel, TorchMixtureModel) else None
        if model_cls in (IonicModel, TorchIonicModel):
            types = (np.arange(len(pos)) % 2).astype(np.int32)    # (equal numbers of the two species: a neutral system)
        sysm = standin.System(pos, L, types=types, dtype=torch.float32, device=dev)
        sysm.randomize_velocities(kT=1.0, seed=7)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(args.dt)
        tfc = htf.tfcompute(model_cls(NN))
        tfc.attach(sim.nlist_cell(r_buff=args.rbuff, check_period=args.check_period), r_cut=rcut)
        sim.run(max(5, args.warmup))
        torch.cuda.synchronize()
        e_warm = float(tfc.force[:, 3].double().sum().item()) / sysm.N   # (compared between routes: same step count here)
        els = []
        for _ in range(3):   # (windows of 15-400 ms: the median of three keeps a one-off stall -- a lazy module load, the
            t0 = time.perf_counter()                                       # run loop's own graph-or-not measurement -- out of the line)
            sim.run(steps)
            torch.cuda.synchronize()
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
        f = tfc.force
        assert bool(torch.isfinite(f).all())
        return {"steps_per_s": steps / el, "ms_per_step": el / steps * 1e3, "particles": sysm.N, "steps": steps,
                "windows_ms_per_step": [e / steps * 1e3 for e in els], "replayed_without_python": tfc._plan is not None,
                "potential_kind": getattr(tfc._plan, "kind", None), "energy_per_particle_after_warmup": e_warm,
                "energy_per_particle": float(f[:, 3].double().sum().item()) / sysm.N}

    class ManyBodyModel(htf.SimModel):
        """Outside the zoo AND not elementwise (round 6): E_i = u_i + 0.02 u_i^2 with u_i the particle's Lennard-Jones energy -- a
        per-row reduction fed into a nonlinearity.  Traced into ONE generated unit with a row function: the one-kernel step."""
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            u = htf.reduce_sum(2.0 * (s ** 12 - s ** 6), axis=1)
            return htf.compute_nlist_forces(nlist, u + 0.02 * u * u)

    class TorchManyBodyModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            u = torch.sum(2.0 * (s ** 12 - s ** 6), dim=1)
            return htf.compute_nlist_forces(nlist, u + 0.02 * u * u)

    class EmbeddedAtomModel(htf.SimModel):
        """Finnis-Sinclair form: -A sqrt(rho_i) + pair repulsion, rho_i = sum_j exp(-1.7 r) / r^2 -- two terms of two different
        sums: two generated units; the plan is the one-kernel step of the first + a streaming evaluation of the second."""
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            rho = htf.reduce_sum(htf.exp(-1.7 * r) * s * s, axis=1)
            phi = htf.reduce_sum(2.0 * s ** 12, axis=1)
            return htf.compute_nlist_forces(nlist, -1.3 * htf.sqrt(rho) + phi)

    class TorchEmbeddedAtomModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            rho = torch.sum(torch.exp(-1.7 * r) * s * s, dim=1)
            # (sqrt'(0) = inf meets the padded slots' zero gradient: a row without neighbors would be NaN in TensorFlow too; none here)
            return htf.compute_nlist_forces(nlist, -1.3 * torch.sqrt(rho) + torch.sum(2.0 * s ** 12, dim=1))

    class TracedLJ(htf.SimModel):
        """examples/06 Force Matching as a user writes it: a Lennard-Jones energy whose prefactor and length are elements of a
        trainable weight vector (build_examples.py:336-372).  The weights are kernel arguments of the generated evaluator and
        of the generated training sweep (forward-mode jets over (r, w_k))."""
        def setup(self, pref, length):
            self.w = torch.nn.Parameter(torch.tensor([pref, length], device=dev))

        def compute(self, nlist, positions, box):
            q = (self.w[1] * htf.nlist_rinv(nlist)) ** 6
            return htf.compute_nlist_forces(nlist, htf.reduce_sum(self.w[0] * 2.0 * (q * q - q), axis=1)), self.w

    class ZooLJ(htf.SimModel):
        def setup(self, pref, length):
            self.lj = htf.LJLayer(pref, length)

        def compute(self, nlist, positions, box):
            e = htf.reduce_sum(self.lj(htf.safe_norm(nlist[:, :, :3], axis=2)), axis=1)
            return htf.compute_nlist_forces(nlist, e), self.lj.w

    def train_one(lattice, cells, model_cls, steps):
        """Force matching on the fly at every MD step: the lowered LJModel drives the run, the trainable model sees the same
        neighbor table, its loss is the mean squared force difference, Adam on the device (training sweep + optimizer kernel)."""
        pos, L, a = (standin.sc_positions if lattice == "sc" else standin.fcc_positions)(cells, 0.8442)
        rng = np.random.default_rng(7)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=torch.float32, device=dev)
        sysm.randomize_velocities(kT=1.0, seed=7)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(args.dt)
        nlist = sim.nlist_cell(r_buff=args.rbuff, check_period=args.check_period)
        lj = htf.tfcompute(LJModel(NN))
        lj.attach(nlist, r_cut=rcut)
        model = model_cls(NN, pref=0.8, length=1.05, output_forces=False)
        model.compile(htf.optimizers.Adam(0.001), loss='MeanSquaredError')
        tfc = htf.tfcompute(model)
        tfc.attach(nlist, train=True, r_cut=rcut)
        tfc.set_reference_forces(lj)
        sim.run(max(5, args.warmup))
        torch.cuda.synchronize()
        els = []
        for _ in range(3):
            t0 = time.perf_counter()
            sim.run(steps)
            torch.cuda.synchronize()
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
        pot = tfc._train_potential
        w = (model.w if model_cls is TracedLJ else model.lj.w).detach().cpu().numpy()
        assert pot is not None and np.isfinite(w).all()
        return {"steps_per_s": steps / el, "ms_per_step": el / steps * 1e3, "particles": sysm.N, "steps": steps,
                "windows_ms_per_step": [e / steps * 1e3 for e in els], "train_potential_kind": int(pot.kind),
                "weights_after": [float(x) for x in w], "loss": float(tfc._opt_state[20])}

    sizes = {}
    for tag, lattice, cells in (("C2 (sc 32^3 = 32768)", "sc", 32), ("C3 (fcc 32^3 x 4 = 131072)", "fcc", 32)):
        fast = one(lattice, cells, LJModel, args.steps)
        gen = one(lattice, cells, TorchLJModel, max(20, args.steps // 10))
        assert abs(fast["energy_per_particle_after_warmup"] - gen["energy_per_particle_after_warmup"]) < 1e-3 * abs(fast["energy_per_particle_after_warmup"]) + 1e-3
        # round 5: models OUTSIDE the zoo written with htf.* ops are traced into generated kernels (HTF_POT_JIT)
        yuk = one(lattice, cells, YukawaLJModel, args.steps)
        yuk_torch = one(lattice, cells, TorchYukawaLJModel, max(20, args.steps // 10))
        morse = one(lattice, cells, MorseModel, args.steps)
        mix = one(lattice, cells, MixtureModel, args.steps)
        mix_torch = one(lattice, cells, TorchMixtureModel, max(20, args.steps // 10))
        ion = one(lattice, cells, IonicModel, args.steps)
        ion_torch = one(lattice, cells, TorchIonicModel, max(20, args.steps // 10))
        assert ion["potential_kind"] == 9
        assert abs(ion["energy_per_particle_after_warmup"] - ion_torch["energy_per_particle_after_warmup"]) < 1e-3 * abs(ion_torch["energy_per_particle_after_warmup"]) + 1e-3
        assert yuk["potential_kind"] == 9 and morse["potential_kind"] == 9 and mix["potential_kind"] == 9
        assert abs(mix["energy_per_particle_after_warmup"] - mix_torch["energy_per_particle_after_warmup"]) < 1e-3 * abs(mix_torch["energy_per_particle_after_warmup"]) + 1e-3
        assert abs(yuk["energy_per_particle_after_warmup"] - yuk_torch["energy_per_particle_after_warmup"]) < 1e-3 * abs(yuk_torch["energy_per_particle_after_warmup"]) + 1e-3
        mb = one(lattice, cells, ManyBodyModel, args.steps)
        mb_torch = one(lattice, cells, TorchManyBodyModel, max(20, args.steps // 10))
        eam = one(lattice, cells, EmbeddedAtomModel, args.steps)
        eam_torch = one(lattice, cells, TorchEmbeddedAtomModel, max(20, args.steps // 10))
        assert mb["potential_kind"] == 9 and eam["potential_kind"] == 9
        for a_, b_ in ((mb, mb_torch), (eam, eam_torch)):
            assert abs(a_["energy_per_particle_after_warmup"] - b_["energy_per_particle_after_warmup"]) < 1e-3 * abs(b_["energy_per_particle_after_warmup"]) + 1e-3
        tr = train_one(lattice, cells, TracedLJ, args.steps)
        tr_zoo = train_one(lattice, cells, ZooLJ, args.steps)
        assert tr["train_potential_kind"] == 9
        assert abs(tr["weights_after"][1] - tr_zoo["weights_after"][1]) < 2e-3   # the same walk as the zoo's LJLayer
        sizes[tag] = {"lowered": fast, "traced_trainable": tr, "zoo_trainable": tr_zoo,
                      "traced_many_body": mb, "torch_many_body": mb_torch, "traced_embedded_atom": eam, "torch_embedded_atom": eam_torch,
                      "many_body_over_lowered_lj_time": mb["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_many_body_time": mb_torch["ms_per_step"] / mb["ms_per_step"],
                      "torch_over_traced_embedded_atom_time": eam_torch["ms_per_step"] / eam["ms_per_step"],
                      "traced_trainable_over_traced_inference_time": tr["ms_per_step"] / yuk["ms_per_step"],
                      "traced_over_zoo_trainable_time": tr["ms_per_step"] / tr_zoo["ms_per_step"], "generic": gen, "generic_over_lowered_time": gen["ms_per_step"] / fast["ms_per_step"],
                      "traced_yukawa_lj": yuk, "traced_morse": morse, "torch_yukawa_lj": yuk_torch,
                      "traced_binary_mixture": mix, "torch_binary_mixture": mix_torch,
                      "traced_ionic": ion, "torch_ionic": ion_torch,
                      "ionic_over_lowered_lj_time": ion["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_ionic_time": ion_torch["ms_per_step"] / ion["ms_per_step"],
                      "mixture_over_lowered_lj_time": mix["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_mixture_time": mix_torch["ms_per_step"] / mix["ms_per_step"],
                      "traced_over_lowered_lj_time": yuk["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_time": yuk_torch["ms_per_step"] / yuk["ms_per_step"]}
    c3 = sizes["C3 (fcc 32^3 x 4 = 131072)"]
    out = {
        "metric": "MD steps/sec, LJModel written in plain torch ops (generic autograd route) at 131072 particles NN=%d" % NN,
        "value": c3["generic"]["steps_per_s"], "unit": "steps/s", "n_gpus": 1, "steps": c3["generic"]["steps"], "warmup": max(5, args.warmup),
        "ms_per_step": c3["generic"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LJModel through tfcompute, htf.* expression layer (lowered) vs plain torch ops + torch.autograd (generic), "
                               "jittered lattices at rho 0.8442, r_cut %.1f, r_buff %.1f, NN %d, dt %g" % (rcut, args.rbuff, NN, args.dt)},
        "sizes": sizes,
        "traced_models": "written with htf.* ops outside the zoo (LJ + Yukawa; a masked Morse well; a Kob-Andersen binary LJ mixture whose "
                         "epsilon / sigma are gathered by species pair from positions[:, 3] and nlist[:, :, 3]; LJ cores + erfc-damped electrostatics between "
                         "two charged species): traced, lowered to generated kernels "
                         "(HTF_POT_JIT: hoomd_tf_amd/codegen.py -> hipcc --genco around csrc/jit_unit.hip), replayed as the one-kernel step",
        "row_functions": "round 6: energies that feed a per-particle reduction into a nonlinearity (traced_many_body: E_i = u_i + 0.02 u_i^2 of the "
                         "particle's LJ energy, one generated unit with a row function, replayed as the one-kernel step; traced_embedded_atom: "
                         "-A sqrt(rho_i) + pair repulsion, two units: the one-kernel step of the first term + a streaming evaluation of the second on the tensor it wrote) against the same models in torch ops + autograd",
        "traced_trainable": "examples/06's trainable Lennard-Jones written with htf.* ops over a weight vector, trained by force matching at "
                            "EVERY MD step while the lowered LJModel drives the run: the step is LJ force kernel + integrator + the generated "
                            "training sweep (weights are kernel arguments; forward-mode jets over (r, w_k)) + the device Adam; zoo_trainable is "
                            "the same with htf.LJLayer (the library's own closed-form sweep)",
        "note": "the generic route keeps the reference's arbitrary-model capability (htf/simmodel.py:87-121, 526-555); models made of "
                "nlist_rinv polynomials, WCARepulsion, RBFExpansion + Dense stacks, EDS biases and compute_rdf are lowered to fused kernels",
        "roofline": None, "cpu_baseline": None,
    }
    print(json.dumps(out))
