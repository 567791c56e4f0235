"""Keras optimizers for the device-side training step (htf_optimizer_step): the update
rules of tf.keras.optimizers.{SGD, Adam, Nadam} (TF 2.3/2.4 optimizer_v2), default
hyper-parameters included."""
from . import _lib


class Optimizer:
    kind = None

    def __init__(self, learning_rate, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon

    def desc(self, nonneg_mask=0, l1_reg=()):
        d = _lib.OptimizerDesc()
        d.kind = self.kind
        d.lr, d.beta1, d.beta2, d.epsilon = self.learning_rate, self.beta_1, self.beta_2, self.epsilon
        d.nonneg_mask = int(nonneg_mask)
        for k, v in enumerate(l1_reg):
            d.l1_reg[k] = float(v)
        return d


    def torch(self, params):
        """The same rule as a torch optimizer: the generic (autograd) training route of SURVEY 8(f)-3."""
        import torch
        if self.kind == _lib.OPT_SGD:
            return torch.optim.SGD(params, lr=self.learning_rate)
        if self.kind == _lib.OPT_ADAM:
            return torch.optim.Adam(params, lr=self.learning_rate, betas=(self.beta_1, self.beta_2), eps=self.epsilon)
        return torch.optim.NAdam(params, lr=self.learning_rate, betas=(self.beta_1, self.beta_2), eps=self.epsilon,
                                 momentum_decay=0.004)


class SGD(Optimizer):
    kind = _lib.OPT_SGD

    def __init__(self, learning_rate=0.01):
        super().__init__(learning_rate)


class Adam(Optimizer):
    kind = _lib.OPT_ADAM

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        super().__init__(learning_rate, beta_1, beta_2, epsilon)


class Nadam(Optimizer):
    kind = _lib.OPT_NADAM

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        super().__init__(learning_rate, beta_1, beta_2, epsilon)


def get(identifier):
    """tf.keras.optimizers.get for the rules built so far."""
    if isinstance(identifier, Optimizer):
        return identifier
    table = {"sgd": SGD, "adam": Adam, "nadam": Nadam}
    try:
        return table[str(identifier).lower()]()
    except KeyError:
        raise ValueError("optimizer %r is not built; available: SGD, Adam, Nadam" % (identifier,))
