"""Configuration tree for the annealed-Langevin channel estimator.

The reference keeps its settings in a ``dotmap.DotMap`` that is built in
``src/score_based_channels/train_score.py:34-67,98-115`` and pickled into the
checkpoint.  ``dotmap`` is not a dependency of this package, so ``Config`` is a
small attribute dictionary with the two behaviours the hot path relies on:

* attribute access auto-creates missing nodes (``config.sampling.steps_each = 3``
  works on a checkpoint config that has no ``sampling`` node,
  ``test_score.py:56``);
* an auto-created, empty node is *falsy* -- this is what makes
  ``config.data.logit_transform`` / ``config.data.rescaled`` evaluate false and
  selects the ``h = 2*x - 1`` input map (``ncsnv2/models/ncsnv2.py:270-273``).
"""
import copy

# Product defaults shared by ScoreNet, the driver, the CLIs and bench.py (kept here: this module imports neither torch nor HIP).
CONV_MODES = ('f16x2', 'bf16x3', 'f32', 'f16w')      # scorenet.ScoreNet(conv_mode=...)
DEFAULT_CONV_MODE = 'f16x2'     # fastest fp32-class multiplier (tests/test_gpu_parity.py::test_f16x2_is_fp32_class)
DEFAULT_STREAMS = 2             # concurrent sub-batch streams of a lock-step chunk (driver.run_concurrently): +5 %, bit-identical


class Config(dict):
    """DotMap-compatible attribute dictionary (auto-vivifying, empty == False)."""

    def __getattr__(self, key):
        if key.startswith('__') and key.endswith('__'):
            raise AttributeError(key)
        if key not in self:
            self[key] = Config()
        return self[key]

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def __bool__(self):
        return len(self) > 0

    def __deepcopy__(self, memo):
        out = Config()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, Config) else v) for k, v in self.items()}

    @staticmethod
    def from_mapping(mapping):
        """Build a Config from any nested mapping (e.g. an unpickled DotMap's dict)."""
        out = Config()
        for k, v in dict(mapping).items():
            out[k] = Config.from_mapping(v) if isinstance(v, dict) else v
        return out


def default_config(channel='CDL-C', image_size=(16, 64), num_classes=2311, ngf=32):
    """The model/data settings ``train_score.py`` writes into ``final_model.pt``.

    Values follow ``train_score.py:34-67`` (model, data) and ``:98-101`` (sigma
    schedule).  ``image_size = [Nr, Nt]``; only ``image_size[1] = Nt`` is read by
    the inference scripts (``test_score.py:75,100``).
    """
    c = Config()
    c.device = 'cuda:0'
    c.model.ema = True
    c.model.ema_rate = 0.999
    c.model.normalization = 'InstanceNorm++'
    c.model.nonlinearity = 'elu'
    c.model.sigma_dist = 'geometric'
    c.model.num_classes = int(num_classes)
    c.model.ngf = int(ngf)
    c.model.sigma_begin = 39.15
    c.model.sigma_rate = 0.995
    c.model.sigma_end = c.model.sigma_begin * c.model.sigma_rate ** (c.model.num_classes - 1)
    c.data.channel = channel
    c.data.channels = 2
    c.data.noise_std = 0
    c.data.image_size = [int(image_size[0]), int(image_size[1])]
    c.data.num_pilots = int(image_size[1])
    c.data.norm_channels = 'global'
    c.data.spacing_list = [0.5]
    return c
