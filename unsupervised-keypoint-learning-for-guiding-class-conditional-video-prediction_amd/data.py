"""Image-pair input pipeline: the reference's ``ImagePairDataLoader`` + ``BaseDataLoader.get_dataset``
(data/image_pair_dataloader.py:16-165, data/base_dataloader.py:33-54, utils/data.py:8-37) re-designed for an MI355X host.

Contract (same as the reference): samples are dicts ``{'image', 'future_image'}`` of 128x128x3 frames of one Penn Action video,
8..11 frames apart (wrapping around), optionally rotated by -10..10 degrees, resized so the short side is 128, cropped
(random offset or centre), flipped and passed through one of ten PIL filters, the SAME random decisions for both frames;
``sample_generator`` yields them scaled to [0,1] (float64, like ``np.asarray(img) / 255.0``) and ``map_fn`` maps to [-1,1].

What is different from the reference's tf.data graph:

* every random decision of a sample is drawn up front into a ``PairPlan`` -- in exactly the order the reference draws them from
  the global ``random`` / ``np.random`` state, so the same seeds reproduce the reference's samples bit for bit
  (tests/golden/image_pair_ref.npz) -- and the pixel work (JPEG decode, rotate, resize, crop, filter: PIL, releases the GIL)
  runs on a thread pool (the reference: ``dataset.map(num_parallel_calls=12)``);
* batches are assembled as **uint8** directly in pinned host memory, copied to HBM asynchronously on a dedicated HIP stream
  (a quarter of the PCIe bytes of float32) and converted to float32 [-1,1] on the GPU by ``kpx_u8_to_unit_f32`` with the
  reference's arithmetic (float32(u8 / 255.0) * 2 - 1), double-buffered so the next batch's copy overlaps the current step
  (the reference: ``dataset.prefetch(1)`` + feed_dict);
* ``shuffle(2000)`` of the reference acts on an i.i.d. stream when ``random_order`` is set (a no-op in distribution); for
  sequential order a plan-level shuffle buffer of the same size is kept, so nothing bigger than indices is buffered.
"""
import os
import os.path as osp
import random as _random
import threading
from collections import deque, namedtuple
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image, ImageEnhance, ImageFilter

IMAGE_SIZE = 128      # reference :13

# utils/data.py:8-35 -- filter ids 0..5 are fixed kernels, 6..9 enhancers with an integer strength drawn from [lo, hi] (x 0.1)
_KERNEL_FILTERS = (ImageFilter.DETAIL, ImageFilter.EDGE_ENHANCE, ImageFilter.SMOOTH, ImageFilter.SMOOTH_MORE,
                   ImageFilter.EDGE_ENHANCE_MORE, ImageFilter.BLUR)
_ENHANCERS = {6: (ImageEnhance.Sharpness, 0, 50), 7: (ImageEnhance.Brightness, 7, 20),
              8: (ImageEnhance.Color, 0, 50), 9: (ImageEnhance.Contrast, 7, 30)}

PairPlan = namedtuple('PairPlan', 'video im_idx fu_idx angle crop flip filt filt_val')


def _resized_size(w, h):
    """Short side -> 128 with the reference's truncations (:107-111, :137-141)."""
    ratio = (h if w > h else w) / float(IMAGE_SIZE)
    return int(w / ratio), int(h / ratio), ratio


def apply_filter(images, filt, filt_val):
    """One of the ten filters of utils/data.py:8-35, the same one for every image of the list."""
    if filt < 6:
        return [im.filter(_KERNEL_FILTERS[filt]) for im in images]
    enhancer = _ENHANCERS[filt][0]
    return [enhancer(im).enhance(filt_val * 0.1) for im in images]


class ImagePairDataLoader(object):
    """Same constructor and methods as the reference class (data/image_pair_dataloader.py:16)."""

    def __init__(self, data_dir, subset, random_order=True, randomness=False, rng=None, np_rng=None):
        self._data_dir = data_dir
        self._random_order = random_order
        self._randomness = randomness
        self._rng = rng if rng is not None else _random            # the reference uses the global generators
        self._np_rng = np_rng if np_rng is not None else np.random
        with open(osp.join(data_dir, subset + '_set.txt'), 'r') as f:
            self._images = f.read().splitlines()
        self._total = len(self._images)
        self._n_files = {}
        print(subset + 'set : ', self._total)

    # ---- reference surface
    def length(self):
        return self._total

    def get_sample_shape(self):
        return {'image': [IMAGE_SIZE, IMAGE_SIZE, 3], 'future_image': [IMAGE_SIZE, IMAGE_SIZE, 3]}

    def get_sample_dtype(self):
        import torch
        return {'image': torch.float32, 'future_image': torch.float32}

    def map_fn(self, inputs):
        """reference :64-69"""
        return {'image': inputs['image'] * 2.0 - 1.0, 'future_image': inputs['future_image'] * 2.0 - 1.0}

    def sample_generator(self):
        """reference :50-62: ``length()`` samples per pass, each {'image','future_image'} in [0,1]."""
        for plan in self.plans():
            im, fu = self.render(plan)
            yield {'image': im / 255.0, 'future_image': fu / 255.0}

    # ---- planning: all random decisions of one sample, in the reference's draw order
    def plans(self):
        if self._random_order:
            for _ in range(self._total):
                yield self.plan(int(self._np_rng.randint(len(self._images))))        # :54
        else:
            for idx in range(self._total):
                yield self.plan(idx)

    def _frame_path(self, video, idx):
        return osp.join(self._data_dir, video, '{:06d}'.format(idx + 1) + '.jpg')

    def plan(self, idx):
        video = self._images[idx].split()[0]
        n_files = self._n_files.get(video)
        if n_files is None:
            n_files = self._n_files[video] = len(os.listdir(osp.join(self._data_dir, video)))
        rng = self._rng
        im_idx, fu_idx = 0, 10                                                       # :75-76
        if self._random_order:
            interval = rng.randint(8, 11)                                            # :79
            im_idx = rng.randint(0, n_files - 1)
            fu_idx = (im_idx + interval) % n_files
        angle = crop = flip = filt = filt_val = None
        if self._randomness:
            angle = rng.randrange(-10, 11)                                           # :93
            with Image.open(self._frame_path(video, im_idx)) as im:                  # header only: the crop range needs the size
                w, h = im.size
            rw, rh, _ = _resized_size(w, h)
            crop = rng.randint(0, int((rw if w > h else rh) - IMAGE_SIZE))            # :113 / :143 (int(w/ratio - 128))
            flip = rng.randint(0, 1)
            filt = rng.randint(0, 9)                                                 # utils/data.py:9
            if filt >= 6:
                filt_val = rng.randint(_ENHANCERS[filt][1], _ENHANCERS[filt][2])
        return PairPlan(video, im_idx, fu_idx, angle, crop, flip, filt, filt_val)

    # ---- pixels (thread-safe: no shared state, PIL releases the GIL in its C loops)
    def render(self, plan):
        """-> (image, future_image) uint8 [128,128,3]"""
        frames = [Image.open(self._frame_path(plan.video, i)) for i in (plan.im_idx, plan.fu_idx)]
        w, h = frames[0].size
        if self._randomness:
            frames = [f.rotate(plan.angle) for f in frames]                          # :94-95
        landscape = w > h
        rw, rh, _ = _resized_size(w, h)
        frames = [f.resize([rw, rh]) for f in frames]
        if self._randomness:
            c = plan.crop
            box = (c, 0, c + IMAGE_SIZE, IMAGE_SIZE) if landscape else (0, c, IMAGE_SIZE, c + IMAGE_SIZE)
            frames = [f.crop(box) for f in frames]
            if plan.flip:
                frames = [f.transpose(Image.FLIP_LEFT_RIGHT) for f in frames]
            frames = apply_filter(frames, plan.filt, plan.filt_val)
        else:
            # the reference centres on the WIDTH in both branches (:126-131 and :156-161): a portrait frame keeps its top 128 rows
            ox = frames[0].size[0] / 2.0
            half = IMAGE_SIZE // 2
            frames = [f.crop((ox - half, 0, ox + half, IMAGE_SIZE)) for f in frames]
        return tuple(np.asarray(f) for f in frames)

    # ---- batches on the device
    def batches(self, batch_size, device, repeat=True, shuffle=False, num_preprocess_threads=8, prefetch=2):
        """The reference's get_dataset(batch_size, repeat, shuffle, num_preprocess_threads, prefetch) as an iterator of
        {'image','future_image'} float32 tensors [B,128,128,3] in [-1,1] living on ``device``.  (The reference maps with 12
        threads; 6-8 decode threads already deliver ~1400 pairs/s of 480x270 JPEGs and leave the GIL to the thread that
        enqueues the train step -- measured with bench_input.py: 890-926 pairs/s fed vs 926 synthetic.)"""
        return _DeviceBatcher(self, batch_size, device, repeat, shuffle, num_preprocess_threads, prefetch)


def _u8_to_unit_numpy(u8):
    """float32(u8 / 255.0) * 2 - 1: sample_generator's float64 division, tf.data's float32 cast, map_fn in float32."""
    return (u8.astype(np.float64) / 255.0).astype(np.float32) * np.float32(2.0) - np.float32(1.0)


class _DeviceBatcher(object):
    SHUFFLE_BUFFER = 2000          # base_dataloader.py:46

    def __init__(self, loader, batch_size, device, repeat, shuffle, threads, prefetch):
        import torch
        self._torch = torch
        self.loader, self.batch_size, self.repeat, self.shuffle = loader, batch_size, repeat, shuffle
        self.device = torch.device(device)
        self.on_gpu = self.device.type == 'cuda'
        self.pool = ThreadPoolExecutor(max_workers=max(1, threads))
        self.depth = max(1, prefetch)
        shape = (2, batch_size, IMAGE_SIZE, IMAGE_SIZE, 3)
        self.host = [torch.empty(shape, dtype=torch.uint8) for _ in range(self.depth + 1)]
        if self.on_gpu:
            self.host = [h.pin_memory() for h in self.host]
            self.copy_stream = torch.cuda.Stream(device=self.device)
        self.copied = [None] * len(self.host)       # per pinned slot: event after its last H2D copy (the slot is refilled later)
        self._plans = self._plan_stream()
        self._inflight = deque()
        self._slot = 0
        self._lock = threading.Lock()

    def _plan_stream(self):
        buf = []
        while True:
            for plan in self.loader.plans():
                if not self.shuffle or self.loader._random_order:
                    yield plan
                else:
                    buf.append(plan)
                    if len(buf) >= self.SHUFFLE_BUFFER:
                        yield buf.pop(self.loader._rng.randrange(len(buf)))
            if not self.repeat:
                break
        while buf:
            yield buf.pop(self.loader._rng.randrange(len(buf)))

    def _fill(self, host_np, k, plan):
        im, fu = self.loader.render(plan)
        host_np[0, k] = im
        host_np[1, k] = fu

    def _submit(self):
        plans = []
        for plan in self._plans:
            plans.append(plan)
            if len(plans) == self.batch_size:
                break
        if not plans:
            return False
        slot = self._slot
        self._slot = (self._slot + 1) % len(self.host)
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()
            self.copied[slot] = None
        host_np = self.host[slot].numpy()
        futures = [self.pool.submit(self._fill, host_np, k, p) for k, p in enumerate(plans)]
        self._inflight.append((slot, len(plans), futures))
        return True

    def __iter__(self):
        return self

    def __next__(self):
        torch = self._torch
        while len(self._inflight) < self.depth and self._submit():
            pass
        if not self._inflight:
            self.pool.shutdown(wait=False)
            raise StopIteration
        slot, n, futures = self._inflight.popleft()
        for f in futures:
            f.result()
        host = self.host[slot][:, :n]
        if not self.on_gpu:
            both = torch.from_numpy(_u8_to_unit_numpy(host.numpy()))
        else:
            from . import ops
            from ._lib import lib, check
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                dev_u8 = host.to(self.device, non_blocking=True)
                self.copied[slot] = torch.cuda.Event(); self.copied[slot].record(self.copy_stream)
                both = torch.empty(host.shape, dtype=torch.float32, device=self.device)
                check(lib.kpx_u8_to_unit_f32(dev_u8.data_ptr(), dev_u8.numel(), both.data_ptr(), self.copy_stream.cuda_stream), 'kpx_u8_to_unit_f32')
            cur.wait_stream(self.copy_stream)              # the consumer's stream sees the batch; the host never blocks
            both.record_stream(cur); dev_u8.record_stream(self.copy_stream)
        while len(self._inflight) < self.depth and self._submit():      # keep the decode pool busy during the step
            pass
        return {'image': both[0], 'future_image': both[1]}


# ------------------------------------------------------------------------------------------------ whole-video and sequence loaders
MIN_IMAGE_SEQ_LEN = 663      # data/keypoint_dataloader.py:13
N_SEQUENCE_LEN = 33          # data/sequence_dataloader.py:14


def center_crop_box(w, h, target_size=IMAGE_SIZE):
    """utils/data.py:40-60: crop box (in the resized frame) and resize ratio for a centred target x target window."""
    half = target_size // 2
    if w > h:
        ratio = h / float(target_size)
        ox = int(w / ratio) / 2.0
        return (ox - half, 0, ox + half, target_size), ratio
    ratio = w / float(target_size)
    oy = int(h / ratio) / 2.0
    return (0, oy - half, target_size, oy + half), ratio


def rotate_keypoints(keypoints, angle, ox=0, oy=0):
    """utils/data.py:63-72 (rotation by -angle degrees about (ox, oy), same operation order -> same rounding)."""
    import math
    c, s = math.cos(math.radians(-angle)), math.sin(math.radians(-angle))
    qx = ox + c * (keypoints[..., 0] - ox) - s * (keypoints[..., 1] - oy)
    qy = oy + s * (keypoints[..., 0] - ox) + c * (keypoints[..., 1] - oy)
    return np.concatenate([np.expand_dims(qx, -1), np.expand_dims(qy, -1)], axis=-1)


def _to_device_unit(u8, device):
    """uint8 host array -> float32 [-1,1] tensor on ``device`` (pinned copy + kpx_u8_to_unit_f32 on a GPU)."""
    import torch
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.from_numpy(_u8_to_unit_numpy(u8))
    from ._lib import lib, check
    host = torch.from_numpy(np.ascontiguousarray(u8)).pin_memory()
    dev_u8 = host.to(device, non_blocking=True)
    out = torch.empty(host.shape, dtype=torch.float32, device=device)
    check(lib.kpx_u8_to_unit_f32(dev_u8.data_ptr(), dev_u8.numel(), out.data_ptr(), torch.cuda.current_stream(device).cuda_stream), 'kpx_u8_to_unit_f32')
    return out


class _ListedVideos(object):
    def __init__(self, data_dir, subset):
        self._data_dir = data_dir
        with open(osp.join(data_dir, subset + '_set.txt'), 'r') as f:
            self._images = f.read().splitlines()
        self._total = len(self._images)
        print(subset + 'set : ', self._total)

    def length(self):
        return self._total

    def _frame(self, video, idx):
        return Image.open(osp.join(self._data_dir, video, '{:06d}'.format(idx + 1) + '.jpg'))

    def _n_frames(self, video):
        return len(os.listdir(osp.join(self._data_dir, video)))


class KeypointDataLoader(_ListedVideos):
    """data/keypoint_dataloader.py:16-86: every frame of a video, centre-cropped to 128x128, zero-padded to 663 frames."""

    def __init__(self, data_dir, subset, threads=8):
        super(KeypointDataLoader, self).__init__(data_dir, subset)
        self._threads = threads

    def get_sample_shape(self):
        return {'image': [MIN_IMAGE_SEQ_LEN, IMAGE_SIZE, IMAGE_SIZE, 3], 'len': None, 'idx': None}

    def get_sample_dtype(self):
        import torch
        return {'image': torch.float32, 'len': torch.int16, 'idx': torch.int16}

    def map_fn(self, inputs):
        return {'image': inputs['image'] * 2.0 - 1.0, 'len': inputs['len'], 'idx': inputs['idx']}

    def render_video(self, idx):
        """-> (uint8 [len,128,128,3], len, video id)"""
        video = self._images[idx].split()[0]
        n = self._n_frames(video)
        with self._frame(video, 0) as first:
            w, h = first.size
        box, ratio = center_crop_box(w, h)
        size = [int(w / ratio), int(h / ratio)]

        def one(i):
            return np.asarray(self._frame(video, i).resize(size).crop(box))
        with ThreadPoolExecutor(max_workers=self._threads) as pool:
            frames = list(pool.map(one, range(n)))
        return np.stack(frames), n, int(video.split('/')[-1])

    def sample_generator(self):
        for idx in range(self._total):
            frames, n, vid = self.render_video(idx)
            seq = frames
            if n < MIN_IMAGE_SEQ_LEN:
                seq = np.concatenate([frames, np.zeros([MIN_IMAGE_SEQ_LEN - n, IMAGE_SIZE, IMAGE_SIZE, 3])], axis=0)
            yield {'image': seq / 255.0, 'idx': vid, 'len': n}

    def videos(self, device):
        """Device-side iterator: {'image': float32 [663,128,128,3] in [-1,1] (pad frames = -1 like the reference), 'len', 'idx'}."""
        for idx in range(self._total):
            frames, n, vid = self.render_video(idx)
            if n < MIN_IMAGE_SEQ_LEN:
                frames = np.concatenate([frames, np.zeros([MIN_IMAGE_SEQ_LEN - n, IMAGE_SIZE, IMAGE_SIZE, 3], dtype=np.uint8)], axis=0)
            yield {'image': _to_device_unit(frames, device), 'len': n, 'idx': vid}


SeqPlan = namedtuple('SeqPlan', 'index video n_files im_idx angle flip scale')


class SequenceDataLoader(_ListedVideos):
    """data/sequence_dataloader.py:17-198: start frame + 32 future key-point sets (+ optionally the future frames) of a video."""

    def __init__(self, data_dir, subset, n_points, n_action, with_image_seq=False, random_order=True, randomness=False,
                 rng=None, np_rng=None):
        super(SequenceDataLoader, self).__init__(data_dir, subset)
        self.n_points, self.n_action = n_points, n_action
        self._with_image_seq, self._random_order, self._randomness = with_image_seq, random_order, randomness
        self._rng = rng if rng is not None else _random
        self._np_rng = np_rng if np_rng is not None else np.random

    def get_sample_shape(self):
        d = {'image': [IMAGE_SIZE, IMAGE_SIZE, 3], 'keypoints': [self.n_points, 2],
             'real_seq': [N_SEQUENCE_LEN - 1, self.n_points, 2], 'action_code': [self.n_action]}
        if self._with_image_seq:
            d['real_im_seq'] = [N_SEQUENCE_LEN - 1, IMAGE_SIZE, IMAGE_SIZE, 3]
        return d

    def get_sample_dtype(self):
        import torch
        return {k: torch.float32 for k in self.get_sample_shape()}

    def map_fn(self, inputs):
        out = {'image': inputs['image'] * 2.0 - 1.0, 'keypoints': inputs['keypoints'], 'real_seq': inputs['real_seq'],
               'action_code': inputs['action_code']}
        if self._with_image_seq:
            out['real_im_seq'] = inputs['real_im_seq'] * 2.0 - 1.0
        return out

    def plans(self):
        if self._random_order:
            for _ in range(self._total):
                yield self.plan(int(self._np_rng.randint(len(self._images))))
        else:
            for idx in range(self._total):
                yield self.plan(idx)

    def plan(self, idx):
        """All random draws of one sample, in the reference's order (:112-118 start frame, :140 rotation, :177 flip, :186 scale)."""
        video = self._images[idx].split()[0]
        n = self._n_frames(video)
        gap = int(n / N_SEQUENCE_LEN)
        im_idx, angle, flip, scale = 0, None, None, None
        if self._randomness:
            rng = self._rng
            im_idx = rng.randint(0, n - N_SEQUENCE_LEN * gap) if gap >= 1 else rng.randint(0, n - ((N_SEQUENCE_LEN - 1) // 2 + 1))
            angle = rng.randrange(-15, 16)
            flip = rng.randint(0, 1)
            scale = rng.randint(70, 120) / 100.0
        return SeqPlan(idx, video, n, im_idx, angle, flip, scale)

    def render(self, plan):
        """-> dict with uint8 'image' (and 'real_im_seq'), float 'keypoints' / 'real_seq' / 'action_code' (before the /255)."""
        video, n, im_idx = plan.video, plan.n_files, plan.im_idx
        action_idx = self._images[plan.index].split()[1]
        keypoints = np.load(osp.join(self._data_dir, video.replace('frames', 'pseudo_labels') + '.npy'))
        gap = int(n / N_SEQUENCE_LEN)
        image = self._frame(video, im_idx)
        if gap >= 1:
            real_seq = keypoints[[im_idx + gap * i for i in range(N_SEQUENCE_LEN)], :, :]
        else:                                    # short video: every second entry is the mean of its neighbours (:126-136)
            n_seq = (N_SEQUENCE_LEN - 1) // 2 + 1
            real_seq = np.zeros([N_SEQUENCE_LEN, self.n_points, 2])
            half = keypoints[im_idx:im_idx + n_seq, :, :]
            for i in range(n_seq - 1):
                real_seq[i * 2] = half[i]
                real_seq[i * 2 + 1] = (half[i] + half[i + 1]) / 2.0
            real_seq[-1] = half[-1]
        if self._randomness:
            image = image.rotate(plan.angle)
            real_seq = rotate_keypoints(real_seq, plan.angle)
        w, h = image.size
        box, ratio = center_crop_box(w, h)
        size = [int(w / ratio), int(h / ratio)]
        image = image.resize(size).crop(box)
        out = {}
        if self._with_image_seq:                 # the future frames start at frame gap (not im_idx + gap) and are never rotated / flipped (:152-171)
            n_future, twice = N_SEQUENCE_LEN - 1, False
            if gap < 1:
                gap, twice, n_future = 1, True, (N_SEQUENCE_LEN - 1) // 2
            seq = []
            for i in range(1, n_future + 1):
                cur = np.asarray(self._frame(video, i * gap).resize(size).crop(box))
                seq.append(cur)
                if twice:
                    seq.append(cur)
            out['real_im_seq'] = np.stack(seq)
        if self._randomness and plan.flip:
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
            real_seq[:, :, 0] *= -1
        label = np.zeros(self.n_action)
        label[int(action_idx)] = 1
        if self._randomness:
            real_seq *= plan.scale
        out.update({'image': np.asarray(image), 'keypoints': real_seq[0, ::], 'real_seq': real_seq[1:, ::], 'action_code': label})
        return out

    def sample_generator(self):
        for plan in self.plans():
            s = self.render(plan)
            s['image'] = s['image'] / 255.0
            if self._with_image_seq:
                s['real_im_seq'] = s['real_im_seq'] / 255.0
            yield s

    def batches(self, batch_size, device, repeat=False, num_preprocess_threads=8):
        """get_dataset(batch_size, ...) + map_fn: float32 tensors on ``device``; images in [-1,1] (converted on the GPU)."""
        import torch
        pool = ThreadPoolExecutor(max_workers=max(1, num_preprocess_threads))
        try:
            while True:
                chunk = []
                for plan in self.plans():
                    chunk.append(plan)
                    if len(chunk) == batch_size:
                        yield self._collate(list(pool.map(self.render, chunk)), device)
                        chunk = []
                if chunk:
                    yield self._collate(list(pool.map(self.render, chunk)), device)
                if not repeat:
                    break
        finally:
            pool.shutdown(wait=False)

    def _collate(self, samples, device):
        import torch
        out = {'image': _to_device_unit(np.stack([s['image'] for s in samples]), device)}
        for k in ('keypoints', 'real_seq', 'action_code'):
            out[k] = torch.from_numpy(np.stack([s[k] for s in samples]).astype(np.float32)).to(device)
        if self._with_image_seq:
            out['real_im_seq'] = _to_device_unit(np.stack([s['real_im_seq'] for s in samples]), device)
        return out
