"""The reference's model zoo (htf/test-py/build_examples.py) written against
hoomd_tf_amd -- same class names, same compute bodies, the TF calls replaced by the
declarative ops of hoomd_tf_amd.simmodel / layers."""
import torch

import hoomd_tf_amd as htf


class SimplePotential(htf.SimModel):
    # build_examples.py:9-22
    def compute(self, nlist, positions):
        return htf.pairwise_unit_forces(nlist)


class BenchmarkPotential(htf.SimModel):
    # build_examples.py:25-30
    def compute(self, nlist):
        rinv = htf.nlist_rinv(nlist)
        energy = rinv
        forces = htf.compute_nlist_forces(nlist, energy)
        return forces


class LJModel(htf.SimModel):
    # build_examples.py:67-77
    def compute(self, nlist, positions, box):
        rinv = htf.nlist_rinv(nlist)
        inv_r6 = rinv**6
        p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
        energy = htf.reduce_sum(p_energy, axis=1)
        forces = htf.compute_nlist_forces(nlist, energy)
        return forces


class LJVirialModel(htf.SimModel):
    # build_examples.py:104-115
    def compute(self, nlist, positions, box):
        rinv = htf.nlist_rinv(nlist)
        inv_r6 = rinv**6
        p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
        energy = htf.reduce_sum(p_energy, axis=1)
        forces_and_virial = htf.compute_nlist_forces(nlist, energy, virial=True)
        return forces_and_virial


class LJRDF(htf.SimModel):
    # build_examples.py:287-304 (LJ through nlist_rinv; the reference's divide_no_nan variant is out of scope)
    def setup(self):
        self.rdfs = []

    def compute(self, nlist, positions, box):
        rinv = htf.nlist_rinv(nlist)
        inv_r6 = rinv**6
        p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
        rdf, rs = htf.compute_rdf(nlist, [3, 5], positions[:, 3])
        _, _ = htf.compute_rdf(nlist, [3, 5])
        self.rdfs.append(rdf)
        forces = htf.compute_nlist_forces(nlist, p_energy)
        return forces, rdf


class LJTypedModel(htf.SimModel):
    # build_examples.py:80-101
    def setup(self):
        self.avg_rdfa = htf.MeanTensor()
        self.avg_rdfb = htf.MeanTensor()

    def compute(self, nlist, positions, box):
        rinv = htf.nlist_rinv(nlist)
        inv_r6 = rinv**6
        p_energy = 1e-10 * (inv_r6 * inv_r6 - inv_r6)
        energy = htf.reduce_sum(p_energy, axis=1)
        forces = htf.compute_nlist_forces(nlist, energy)
        rdfa, rs = htf.compute_rdf(nlist, [0, 10], positions[:, 3], type_i=0, type_j=1)
        rdfb, rs = htf.compute_rdf(nlist, [0, 10], positions[:, 3], type_i=1, type_j=0)
        self.avg_rdfa.update_state(rdfa)
        self.avg_rdfb.update_state(rdfb)
        return forces


class WCA(htf.SimModel):
    # build_examples.py:221-228
    def setup(self):
        self.wca = htf.WCARepulsion(0.5)

    def compute(self, nlist):
        energy = self.wca(nlist)
        forces = htf.compute_nlist_forces(nlist, energy)
        return forces


class PairMLPModel(htf.SimModel):
    # RBFExpansion + 2x64 MLP pair potential
    def setup(self, activation='tanh', seed=3, precision='fp32'):
        self.mlp = htf.PairMLP(32, 64, 64, 0.0, 3.0, activation=activation, seed=seed, precision=precision)

    def compute(self, nlist):
        energy = self.mlp(nlist)
        return htf.compute_nlist_forces(nlist, energy)


class WrapModel(htf.SimModel):
    # build_examples.py:49-56
    def compute(self, nlist, positions, box):
        p1 = positions[0, :3]
        p2 = positions[-1, :3]
        r = p1 - p2
        rwrap = htf.wrap_vector(r, box)
        return rwrap


class EDSModel(htf.SimModel):
    # build_examples.py:118-135
    def setup(self, set_point):
        self.cv_sum, self.cv_n = 0.0, 0
        self.eds_bias = htf.EDSLayer(set_point, 5, 1 / 5)

    def compute(self, nlist, positions, box):
        rvec = htf.wrap_vector(positions[0, :3], box)
        cv = torch.sqrt((rvec * rvec).sum())
        self.cv_sum += float(cv)
        self.cv_n += 1
        alpha = self.eds_bias(cv)
        energy = (cv - 5) ** 2 + cv * alpha.detach()
        forces = htf.compute_positions_forces(positions, energy)
        return forces, alpha


class EDSRDFModel(htf.SimModel):
    # config C4: LJModel + EDS bias on a soft RDF collective variable, with the
    # hard compute_rdf as an observable
    def setup(self, set_point, r0=1.1, gap=0.05, period=25, learning_rate=5.0):
        self.soft_bin = htf.SoftRDFCV(r0, gap)
        self.eds_bias = htf.EDSLayer(set_point, period, learning_rate)

    def compute(self, nlist, positions, box):
        rinv = htf.nlist_rinv(nlist)
        inv_r6 = rinv**6
        lj_energy = htf.reduce_sum(4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6), axis=1)
        cv = self.soft_bin(nlist)
        alpha = self.eds_bias(cv)
        energy = lj_energy + alpha * cv
        forces = htf.compute_nlist_forces(nlist, energy)
        rdf, rs = htf.compute_rdf(nlist, [0, 3.5])
        return forces, cv, alpha, rdf


class TrainableGraph(htf.SimModel):
    # build_examples.py:362-372 + example 06 TrainableLJ
    def setup(self, sig=1.0, eps=1.0):
        self.lj = htf.LJLayer(sig, eps)

    def get_layer(self, name):
        return {'lj': self.lj}[name]

    def compute(self, nlist, positions, box):
        r = htf.safe_norm(nlist[:, :, :3], axis=2)
        p_energy = self.lj(r)
        energy = htf.reduce_sum(p_energy, axis=1)
        forces = htf.compute_nlist_forces(nlist, energy)
        return forces, self.lj.w, energy


# ---- generic (torch autograd) models: SURVEY 8(f)-3 --------------------------------------
class LJMolModel(htf.MolSimModel):
    # build_examples.py:310-318, TF calls -> torch
    def mol_compute(self, nlist, positions, mol_nlist, mol_positions, box):
        r = torch.norm(mol_nlist, dim=3)
        rinv = torch.where(r > 0, 1.0 / torch.where(r > 0, r, torch.ones_like(r)), torch.zeros_like(r))  # divide_no_nan
        mol_p_energy = 4.0 / 2.0 * (rinv**12 - rinv**6)
        total_e = torch.sum(mol_p_energy)
        forces = htf.compute_nlist_forces(nlist, total_e)
        return forces


class MappedNlist(htf.SimModel):
    # build_examples.py:183-196
    def my_map(pos, box):
        x = torch.mean(pos[:, :3], dim=0, keepdim=True)
        cg1 = torch.cat((x, torch.zeros((1, 1), dtype=x.dtype, device=x.device)), -1)
        cg2 = torch.tensor([[0, 0, 0.1, 1]], dtype=x.dtype, device=x.device)
        return torch.cat((cg1, cg2), dim=0)

    def compute(self, nlist, positions, box):
        r = torch.norm(nlist[:, :, :3], dim=2)
        nlist, cnlist = self.mapped_nlist(nlist)
        return positions, nlist, cnlist


class TorchLJModel(htf.SimModel):
    # LJModel written in plain torch ops on the neighbor tensor: the generic autograd route
    def compute(self, nlist, positions, box):
        # nlist_rinv op for op (simmodel.py:618-635)
        delta = 3e-6
        r = torch.sqrt(torch.sum((nlist[:, :, :3] + delta / 3 / 10) ** 2, dim=2))
        rinv = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
        inv_r6 = rinv ** 6
        p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
        energy = torch.sum(p_energy, dim=1)
        return htf.compute_nlist_forces(nlist, energy, virial=self.virial)


class NlistNN(htf.SimModel):
    # build_examples.py:199-218
    def setup(self, dim, top_neighs):
        self.dense1 = htf.Dense(dim, seed=1)
        self.dense2 = htf.Dense(dim, seed=2)
        self.last = htf.Dense(1, seed=3)
        self.top_neighs = top_neighs

    def compute(self, nlist, positions, box):
        rinv = htf.nlist_rinv(nlist)
        top_n = htf.sort(rinv, axis=1, direction='DESCENDING')[:, :self.top_neighs]
        x = self.dense1(top_n)
        x = self.dense2(x)
        energy = self.last(x)
        forces = htf.compute_nlist_forces(nlist, energy)
        return forces


class NoForceModel(htf.SimModel):
    # build_examples.py:33-40
    def compute(self, nlist, positions):
        neighs_rs = torch.norm(nlist[:, :, :3], dim=2)
        energy = torch.where(neighs_rs > 0, 1.0 / torch.where(neighs_rs > 0, neighs_rs, torch.ones_like(neighs_rs)),
                             torch.zeros_like(neighs_rs))
        pos_norm = torch.norm(positions, dim=1)
        return energy, pos_norm


class TrainModel(htf.SimModel):
    # build_examples.py:244-268: Dense layers on the sorted 1/r of the closest neighbors; the
    # `training` flag doubles the energy; `output_zero` zeroes the extra output
    def setup(self, dim, top_neighs):
        g = torch.Generator().manual_seed(1)
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.w1 = (0.3 * torch.randn((top_neighs, dim), generator=g)).to(dev).requires_grad_(True)
        self.w2 = (0.3 * torch.randn((dim, dim), generator=g)).to(dev).requires_grad_(True)
        self.w3 = (0.3 * torch.randn((dim, 1), generator=g)).to(dev).requires_grad_(True)
        self.top_neighs = top_neighs
        self.output_zero = False

    def compute(self, nlist, positions, training):
        r = torch.sqrt(torch.sum((nlist[:, :, :3] + 1e-7 / 3) ** 2, dim=2))
        rinv = torch.where(r > 3e-6, 1.0 / (r + 3e-6), torch.zeros_like(r))
        top_n = torch.sort(rinv, dim=1, descending=True)[0][:, :self.top_neighs]
        x = top_n.to(self.w1.dtype) @ self.w1
        x = x @ self.w2
        energy = x @ self.w3
        if training:
            energy = energy * 2
        forces = htf.compute_nlist_forces(nlist, energy)
        if self.output_zero:
            energy = energy * 0.
        return forces, energy


class TensorSaveModel(htf.SimModel):
    # build_examples.py:43-46
    def compute(self, nlist, positions):
        pos_norm = torch.norm(positions, dim=1)
        return pos_norm


class BenchmarkNonlistModel(htf.SimModel):
    # build_examples.py:59-64
    def compute(self, nlist, positions, box):
        ps = htf.norm(positions, axis=1)
        energy = htf.divide_no_nan(1., ps)
        forces = htf.compute_positions_forces(positions, energy)
        return forces


class QuickstartWCA(htf.SimModel):
    # examples/01. Quickstart.ipynb cell 3 (WCAPotential): r^-12 inside 2^(1/6) via a cast mask, an RDF averaged every step
    def setup(self):
        self.avg_rdf = htf.MeanTensor()  # tf.keras.metrics.MeanTensor in the notebook

    def compute(self, nlist):
        r12 = htf.nlist_rinv(nlist)**12
        r = htf.norm(nlist[:, :, :3], axis=2)
        pair_energy = htf.cast(r < 2**(1 / 6), torch.float32) * r12
        particle_energy = htf.reduce_sum(pair_energy, axis=1)
        forces = htf.compute_nlist_forces(nlist, particle_energy)
        inst_rdf = htf.compute_rdf(nlist, [0, 3.5])
        self.avg_rdf.update_state(inst_rdf)
        return forces


class _Mean:
    """tf.keras.metrics.Mean"""

    def __init__(self):
        self.total, self.count = 0.0, 0

    def update_state(self, x):
        x = x.detach().double()
        self.total += float(x.sum())
        self.count += x.numel()

    def result(self):
        return self.total / max(self.count, 1)


class LJRunningMeanModel(htf.SimModel):
    # build_examples.py:270-286 (plain norm + divide_no_nan: the generic route)
    def setup(self):
        self.avg_energy = _Mean()

    def compute(self, nlist, positions, box):
        r = torch.norm(nlist[:, :, :3], dim=2)
        inv_r6 = torch.where(r > 0, 1.0 / torch.where(r > 0, r, torch.ones_like(r)) ** 6, torch.zeros_like(r))
        p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
        energy = torch.sum(p_energy, dim=1)
        self.avg_energy.update_state(energy)
        forces = htf.compute_nlist_forces(nlist, energy)
        return forces
