"""Optimizer objects for Model.compile — the reference's notebook compiles with
`Adam(lr=7e-4, epsilon=1e-8, decay=1e-6)` (segmentation.ipynb cell 2, `from keras.optimizers import Adam`).

The update itself is one launch of libdl3.so over the flat parameter arena (dl3_adam_step, engine.Engine.adam); this
class only carries the hyper-parameters with Keras 2.2.4's names and defaults (keras/optimizers.py `Adam.__init__`:
lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None -> K.epsilon()=1e-7, decay=0., amsgrad=False)."""

KERAS_EPSILON = 1e-7  # keras.backend.epsilon() default


class Adam:
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=None, decay=0.0, amsgrad=False, **kwargs):
        if "learning_rate" in kwargs:  # tf.keras spelling
            lr = kwargs.pop("learning_rate")
        if kwargs:
            raise TypeError("Adam: unexpected keyword arguments %s (clipnorm / clipvalue are not on the path)" % sorted(kwargs))
        if amsgrad:
            raise ValueError("Adam(amsgrad=True) is not implemented by dl3_adam_step (the reference never uses it)")
        if not (lr >= 0 and 0 <= beta_1 < 1 and 0 <= beta_2 < 1 and decay >= 0):
            raise ValueError("Adam: lr, decay must be >= 0 and beta_1, beta_2 in [0, 1)")
        self.lr, self.beta_1, self.beta_2 = float(lr), float(beta_1), float(beta_2)
        self.epsilon = KERAS_EPSILON if epsilon is None else float(epsilon)
        self.decay = float(decay)
        self.amsgrad = False

    def get_config(self):
        return dict(lr=self.lr, beta_1=self.beta_1, beta_2=self.beta_2, epsilon=self.epsilon, decay=self.decay,
                    amsgrad=self.amsgrad)

    @classmethod
    def from_config(cls, cfg):
        return cls(**cfg)

    def __repr__(self):
        return "Adam(%s)" % ", ".join("%s=%g" % kv for kv in self.get_config().items() if kv[0] != "amsgrad")


_KEYS = ("lr", "beta_1", "beta_2", "epsilon", "decay")


def as_adam_dict(optimizer):
    """what Model.compile accepts -> the hyper-parameter dict Engine.adam consumes.
    None: the notebook's values (Engine.adam defaults); dict: partial override of those; 'adam' / Adam(): Keras defaults
    for everything not given; any other object exposing Keras' get_config() with an Adam-shaped config is accepted too."""
    if optimizer is None:
        return {}
    if isinstance(optimizer, dict):
        bad = sorted(set(optimizer) - set(_KEYS))
        if bad:
            raise ValueError("compile(optimizer=dict): unknown keys %s (known: %s)" % (bad, list(_KEYS)))
        return {k: float(v) for k, v in optimizer.items()}
    if isinstance(optimizer, str):
        if optimizer.lower() != "adam":
            raise ValueError("compile(optimizer=%r): only Adam is implemented (the reference trains with Adam)" % optimizer)
        optimizer = Adam()
    cfg = getattr(optimizer, "get_config", None)
    if cfg is None:
        raise TypeError("compile(optimizer=%r): expected None, a dict, 'adam' or an Adam object" % (optimizer,))
    cfg = cfg()
    if cfg.get("amsgrad"):
        raise ValueError("Adam(amsgrad=True) is not implemented by dl3_adam_step")
    name = type(optimizer).__name__.lower()
    if "adam" not in name or "adamax" in name or "nadam" in name:
        raise ValueError("compile(optimizer=%s): only Adam is implemented" % type(optimizer).__name__)
    if "learning_rate" in cfg and "lr" not in cfg:
        cfg["lr"] = cfg["learning_rate"]
    out = {k: float(cfg[k]) for k in _KEYS if cfg.get(k) is not None}
    out.setdefault("epsilon", KERAS_EPSILON)
    out.setdefault("decay", 0.0)
    return out
