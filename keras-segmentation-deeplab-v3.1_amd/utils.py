"""Host-side mirror of the parts of the reference's utils.py that sit on the hot path:
`SegModel.create_seg_model` head surgery (utils.py:169-214), the loss
`sparse_crossentropy_ignoring_last_label` (utils.py:127-130) — executed on the GPU by
dl3_softmax_xent — and host restatements of the metrics that the north star leaves on the host
(`Jaccard` utils.py:139-157, `sparse_accuracy_ignoring_last_label` utils.py:132-138).
Next-ring rows of SURVEY §8(f): `prepare_targets` (N2, the label half of SegmentationGenerator.__getitem__ on the
device) and `Jaccard_from_counts` / `accuracy_from_counts` (N3, metrics from dl3_seg_counts).
`do_crf` (N4) is the host hook with the reference's parameters; it needs the optional pydensecrf package.
Out of scope by the SURVEY §8 contract: image file I/O + cv2 augmentation, plotting.
"""
import numpy as np

from . import graph as G
from .deeplabv3p import Deeplabv3
from .graph import Activation, Conv2D, Model, Reshape, ResizeBilinear
from .subpixel import Subpixel, icnr_weights


def sparse_crossentropy_ignoring_last_label(y_true, y_pred):
    """Marker for Model.compile(loss=...): the engine's training step always evaluates this loss
    (utils.py:127-130) on the GPU; calling it on host arrays evaluates the same formula in numpy."""
    y_true = np.asarray(y_true)
    p = np.asarray(y_pred, np.float64)
    C = p.shape[-1]
    t = y_true[:, :, 0].astype(np.int64)
    q = np.clip(p / p.sum(-1, keepdims=True), 1e-7, 1 - 1e-7)
    valid = t < C
    out = np.zeros(t.shape)
    b, i = np.nonzero(valid)
    out[b, i] = -np.log(q[b, i, t[b, i]])
    return out


def sparse_accuracy_ignoring_last_label(y_true, y_pred):
    """utils.py:132-138 on host."""
    C = y_pred.shape[-1]
    t = np.asarray(y_true).reshape(-1).astype(np.int64)
    pred = np.asarray(y_pred).reshape(-1, C).argmax(-1)
    legal = t != C
    return float((legal & (t == pred)).sum() / max(legal.sum(), 1))


def Jaccard(y_true, y_pred):
    """utils.py:139-157 on host: per class, IoU per image averaged over images containing the class;
    classes that occur nowhere (NaN) are dropped; mean over the remaining classes."""
    y_true = np.asarray(y_true)
    C = y_pred.shape[-1]
    pred = np.asarray(y_pred).argmax(-1)
    t = y_true[:, :, 0]
    ious = []
    for i in range(C):
        tl, pl = t == i, pred == i
        legal = tl.sum(axis=1) > 0
        if legal.any():
            inter = (tl & pl).sum(axis=1)[legal]
            union = (tl | pl).sum(axis=1)[legal]
            ious.append(float(np.mean(inter / union)))
    return float(np.mean(ious)) if ious else float("nan")


def Jaccard_from_counts(counts):
    """`Jaccard` (utils.py:139-157) from the integer counts of dl3_seg_counts, counts[B][3][C] =
    (#true==c, #pred==c, #both): union = true + pred - inter; same float64 ratios and means as `Jaccard`."""
    counts = np.asarray(counts, np.int64)
    t, p, inter = counts[:, 0], counts[:, 1], counts[:, 2]
    ious = []
    for i in range(counts.shape[2]):
        legal = t[:, i] > 0
        if legal.any():
            union = (t[:, i] + p[:, i] - inter[:, i])[legal]
            ious.append(float(np.mean(inter[:, i][legal] / union)))
    return float(np.mean(ious)) if ious else float("nan")


def accuracy_from_counts(counts):
    """`sparse_accuracy_ignoring_last_label` (utils.py:132-138) from dl3_seg_counts output."""
    counts = np.asarray(counts, np.int64)
    return float(counts[:, 2].sum() / max(counts[:, 0].sum(), 1))


def prepare_targets(labels, n_classes=21):
    """Device-side label half of SegmentationGenerator.__getitem__ (utils.py:375-402): raw label maps
    [B,H,W] or [B,HW] (uint8 / int32; numpy array or cuda tensor) -> (Y [B,HW,1], SW [B,HW]) cuda float32 tensors, ready
    for `Model.train_on_batch(X, Y, SW)`.  Runs dl3_prepare_targets; raises if libdl3.so is missing."""
    import torch
    from . import capi
    if isinstance(labels, np.ndarray):
        if labels.dtype not in (np.uint8, np.int32):
            labels = labels.astype(np.int32)
        labels = torch.from_numpy(np.ascontiguousarray(labels)).cuda()
    if labels.dtype not in (torch.uint8, torch.int32):
        labels = labels.to(torch.int32)
    labels = labels.contiguous().reshape(labels.shape[0], -1)
    B, HW = labels.shape
    Y = torch.empty(B, HW, 1, device=labels.device, dtype=torch.float32)
    SW = torch.empty(B, HW, device=labels.device, dtype=torch.float32)
    hist = torch.empty(B, n_classes + 1, device=labels.device, dtype=torch.int32)
    capi.call("dl3_prepare_targets", capi.ptr(labels), capi.LABEL_U8 if labels.dtype == torch.uint8 else capi.LABEL_I32,
              B, HW, n_classes, capi.ptr(Y), capi.ptr(SW), capi.ptr(hist), torch.cuda.current_stream().cuda_stream)
    return Y, SW


# Dense-CRF post-processing stays on the host (north star): the hook and the reference's parameter set
# (utils.py:74-91).  pydensecrf is not a dependency of this package: the hook imports it on first use.
CRF_PARAMS = dict(gt_prob=0.7, gaussian_sxy=(3, 3), gaussian_compat=3, bilateral_sxy=80, bilateral_srgb=13,
                  bilateral_compat=10, iterations=5)


def restore_crf_labels(MAP, colors):
    """MAP indices -> the mask's original values, exactly as the reference does it (utils.py:86-89): for every index u
    present in MAP, ascending, `np.putmask(MAP, MAP == u, colors[u])` IN PLACE.  The quirk is reproduced, not fixed
    (SURVEY G5): a value written for an earlier index is matched again by a later one — with mask values {0, 2, 15} index
    1 becomes 2 and is then rewritten to 15 together with index 2, so class 2 vanishes from the result.  Masks whose
    values are 0..n-1 (the usual label maps) come back unchanged."""
    for u in np.unique(MAP):
        np.putmask(MAP, MAP == u, colors[u])
    return MAP


def do_crf(im, mask, zero_unsure=True):
    """Fully connected CRF refinement of a label mask given the image (reference utils.py:74-91): unary energies from
    the labels with CRF_PARAMS['gt_prob'], a Gaussian (position) and a bilateral (position + colour) pairwise term,
    five mean-field iterations, MAP labels mapped back to the mask's original values by the reference's own in-place
    loop (restore_crf_labels)."""
    try:
        import pydensecrf.densecrf as dcrf
        from pydensecrf.utils import unary_from_labels
    except ImportError as e:  # pragma: no cover - optional host dependency
        raise ImportError("do_crf needs the optional host package pydensecrf (not installed)") from e
    colors, labels = np.unique(mask, return_inverse=True)
    labels = labels.reshape(-1)  # numpy 1.x (the reference's) returns the inverse flat; numpy >= 2 in the mask's shape
    image_size = mask.shape[:2]
    n_labels = len(set(labels.flat))
    crf = dcrf.DenseCRF2D(image_size[1], image_size[0], n_labels)  # width, height, nlabels
    crf.setUnaryEnergy(unary_from_labels(labels, n_labels, gt_prob=CRF_PARAMS["gt_prob"], zero_unsure=zero_unsure))
    crf.addPairwiseGaussian(sxy=CRF_PARAMS["gaussian_sxy"], compat=CRF_PARAMS["gaussian_compat"])
    crf.addPairwiseBilateral(sxy=CRF_PARAMS["bilateral_sxy"], srgb=CRF_PARAMS["bilateral_srgb"],
                             rgbim=np.ascontiguousarray(im.astype("uint8")), compat=CRF_PARAMS["bilateral_compat"])
    q = crf.inference(CRF_PARAMS["iterations"])
    MAP = np.argmax(q, axis=0).reshape(image_size)
    return restore_crf_labels(MAP, colors)


class SegmentationGenerator:
    """The tensor contract of the reference's `SegmentationGenerator` (utils.py:257-409) over IN-MEMORY arrays:
    `gen[i]` -> `(X [B,H,W,3] float32, Y [B,HW,1], {'pred_mask': SW [B,HW]})`, `len(gen)` batches, `on_epoch_end()`
    reshuffles.  File I/O and the cv2 augmentation chain of the reference (utils.py:314-369) are out of scope: the caller
    hands over decoded images (raw 0-255, as `self.X[n] = image`, utils.py:387) and raw label maps; the label half
    (utils.py:371-400) runs on the device through dl3_prepare_targets, so Y and SW are cuda tensors that
    `Model.fit_generator` / `train_on_batch` consume without a host round trip."""

    def __init__(self, images, labels, n_classes=21, batch_size=1, seed=7, shuffle=True):
        self.images = np.asarray(images)
        self.labels = np.asarray(labels)
        if self.images.ndim != 4 or self.images.shape[-1] != 3 or len(self.images) != len(self.labels):
            raise Exception("images must be [N,H,W,3] and labels [N,H,W]")
        self.n_classes = int(n_classes)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self._rng = np.random.RandomState(seed)
        self.order = np.arange(len(self.images))

    def __len__(self):
        return len(self.images) // self.batch_size

    def __getitem__(self, i):
        if not 0 <= i < len(self):
            raise IndexError(i)
        idx = self.order[i * self.batch_size:(i + 1) * self.batch_size]
        X = np.ascontiguousarray(self.images[idx], dtype=np.float32)
        Y, SW = prepare_targets(self.labels[idx], self.n_classes)
        return X, Y, {"pred_mask": SW}

    def on_epoch_end(self):
        if self.shuffle:
            self._rng.shuffle(self.order)


class SegModel:
    """utils.py:160-254 — only model construction is on the path."""
    epochs = 20
    batch_size = 16

    def __init__(self, dataset="VOCdevkit/VOC2012", image_size=(320, 320)):
        self.sz = tuple(image_size)
        self.mainpath = dataset
        self.crop = False

    def create_seg_model(self, net, n=21, backbone="mobilenetv2", load_weights=False, multi_gpu=False):
        """net='original': DeepLabV3+ body + conv_upsample 1x1 + bilinear (utils.py:188-193);
        net='subpixel': body + Subpixel(n, 1, scale) with ICNR init (utils.py:194-204).
        The body is Deeplabv3(weights=None, classes=21, OS=16) cut at model.layers[-5] (utils.py:177-181).
        multi_gpu: the reference's in-graph keras.utils.multi_gpu_model (utils.py:209-211) becomes one process per
        GPU + one RCCL all-reduce of the gradients per step: Model.distribute() attaches parallel.DataParallel, which
        reads RANK / WORLD_SIZE (launch with `python -m torch.distributed.run --nproc-per-node <gpus> ...`); every
        train_on_batch then takes the global batch and trains on this rank's shard."""
        model = Deeplabv3(weights=None, input_tensor=None, infer=False, input_shape=self.sz + (3,), classes=21,
                          backbone=backbone, OS=16, alpha=1)
        base_model = Model(model.input, model.layers[-5].output)
        self.net = net
        self.modelpath = "weights/{}_{}.h5".format(backbone, net)
        scale = 4 if backbone == "xception" else 8
        if net == "original":
            x = Conv2D(n, (1, 1), padding="same", name="conv_upsample")(base_model.output)
            x = ResizeBilinear((self.sz[0], self.sz[1]))(x)
            x = Reshape((self.sz[0] * self.sz[1], -1))(x)
            x = Activation("softmax", name="pred_mask")(x)
            model = Model(base_model.input, x, name="deeplabv3p")
        elif net == "subpixel":
            x = Subpixel(n, 1, scale, padding="same")(base_model.output)
            x = Reshape((self.sz[0] * self.sz[1], -1))(x)
            x = Activation("softmax", name="pred_mask")(x)
            model = Model(base_model.input, x, name="deeplabv3p_subpixel")
        else:
            raise ValueError("net must be 'original' or 'subpixel'")
        for layer in model.layers:  # ICNR re-initialisation (utils.py:200-204)
            if type(layer) == Subpixel:
                c, b = layer.get_weights()
                layer.set_weights([icnr_weights(scale=scale, shape=c.shape), b])
        if load_weights:
            model.load_weights(self.modelpath)
        if multi_gpu:
            model.distribute()
            if model._dp.world == 1:
                import warnings
                warnings.warn("multi_gpu=True in a single process: this package runs one process per GPU — launch the "
                              "script with `python -m torch.distributed.run --nproc-per-node <gpus>` to use them",
                              RuntimeWarning, stacklevel=2)
        self.model = model
        return model

    def load_weights(self, model):
        model.load_weights(self.modelpath)

    @classmethod
    def set_num_epochs(cls, new_epochs):
        cls.epochs = new_epochs

    @classmethod
    def set_batch_size(cls, new_batch_size):
        cls.batch_size = new_batch_size
