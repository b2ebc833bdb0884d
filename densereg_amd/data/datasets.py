"""The three dataset adapters of the reference (``data/dataset_base.py``, ``data/icvl.py``, ``data/nyu.py``,
``data/msra.py``) without TensorFlow / OpenCV: annotation parsers, raw-frame readers, TFRecord shard writer and
reader, and the batch iterator that ends where the network begins.

Per batch: records -> ``parse_example`` (``tfrecord.py``) -> PNG inflate + row filters on the host (``png.py``,
``dr_png_unfilter``) -> the packed samples are uploaded as bytes -> ``dr_depth_from_samples`` (fp32 depth frames on
the device) -> ``dr_crop_from_pose`` / ``dr_crop_from_bbx`` (crop, resize, threshold, crop camera, centre of mass:
``preprocess_op`` of the reference) -> ``[dm (B,128,128,1), pose (B,3J), cfg (B,6), com (B,3), names]``, the tuple
``JointDetectionModel.loss`` / ``.test`` consume.  The queue-runner machinery of the reference (shuffle queues,
reader threads: ``dataset_base.py:153-205``) is replaced by a shard-level and buffer-level shuffle with a seeded
generator, a decode thread pool and a producer thread that keeps two decoded batches ahead of the GPU
(``host_batches``); ranks of a data-parallel job read disjoint shards.

Where the reference is ambiguous this file says what it does:
* ``IcvlDataset.is_train`` returns True for every subset (``icvl.py:46-47``), so its ``loadAnnotation`` would drop
  every test label (names there do not start with '2014').  Here the '2014' filter applies to the training subsets only.
* ``MsraDataset.filenames`` for training loops over the eight other subjects but formats the held-out ``self.pid``
  into every name (``msra.py:52-55``), i.e. it would train on the test subject eight times.  Here: the other subjects.
* ``filenames`` lists the reference's fixed shard names (incl. the repeated last shard, which a reader stopping after
  ``exact_num`` frames never reaches; NYU training reads only shards 0..99 of 300, ``nyu.py:61-64``).  ``files=`` /
  ``files_override`` replace the list for data laid out differently (the tests' miniature datasets).
* NYU keeps 14 of the 36 joints (``nyu.py:40-45``) and flips y (``:119,130``); MSRA flips y and z (``msra.py:103-107``).
"""
from __future__ import annotations

import glob
import os
import pickle
import struct
from collections import namedtuple
from typing import Iterator, List, Optional, Sequence

import numpy as np

from . import png, tfrecord

Annotation = namedtuple('Annotation', 'name,pose,bbx', defaults=(None,))
CameraConfig = namedtuple('CameraConfig', 'fx,fy,cx,cy,w,h')


def uvd2xyz(uvd: np.ndarray, cfg) -> np.ndarray:
    """data/util.py:20-21,41-49: back-projection of (u, v, d) rows."""
    p = np.asarray(uvd, np.float64).reshape(-1, 3)
    return np.stack([(p[:, 0] - cfg[2]) * p[:, 2] / cfg[0], (p[:, 1] - cfg[3]) * p[:, 2] / cfg[1], p[:, 2]], 1)


def xyz2uvd(xyz: np.ndarray, cfg) -> np.ndarray:
    p = np.asarray(xyz, np.float64).reshape(-1, 3)
    return np.stack([p[:, 0] * cfg[0] / p[:, 2] + cfg[2], p[:, 1] * cfg[1] / p[:, 2] + cfg[3], p[:, 2]], 1)


def read_msra_bin(path: str, prev: Optional[np.ndarray] = None) -> np.ndarray:
    """One frame of MSRA15 (``msra.py:120-141``): 6 int32 (cols, rows, left, top, right, bottom) + the cropped float32
    depth; expanded to the full frame; an empty frame repeats the previous one."""
    with open(path, 'rb') as f:
        cols, rows, left, top, right, bottom = struct.unpack('<6i', f.read(24))
        crop = np.fromfile(f, dtype='<f4')
    crop = crop.reshape(bottom - top, right - left)
    dm = np.zeros((rows, cols), np.float32)
    dm[top:bottom, left:right] = crop
    if dm.sum() < 10 and prev is not None:
        dm = prev
    return dm


class BaseDataset(object):
    """``data/dataset_base.py:BaseDataset``."""
    name = 'base'
    cfg = CameraConfig(1, 1, 0, 0, 1, 1)
    approximate_num_per_file = 1
    png_channels, png_depth = 1, 16
    orig_pose_dim = pose_dim = 0
    keep_pose_idx: Optional[np.ndarray] = None

    def __init__(self, subset: str, directory: Optional[str] = None):
        self.subset = subset
        if directory is not None:
            self.directory = directory
        self._annotations: List[Annotation] = []
        self._iter = None
        self.files_override: Optional[List[str]] = None
        self.rank, self.world, self.seed = 0, 1, 0              # data-parallel placement of this reader (set by the driver)

    # -- conversion (dataset_base.py:49-127) ---------------------------------------------------------
    @property
    def annotations(self):
        return self._annotations

    def image_bytes(self, label: Annotation) -> bytes:
        with open(os.path.join(self.img_dir, label.name), 'rb') as f:
            return f.read()

    def convert_to_example(self, label: Annotation) -> bytes:
        feats = {'name': label.name.encode(), 'xyz_pose': np.asarray(label.pose, np.float32), 'png16': self.image_bytes(label)}
        if label.bbx is not None:
            feats['bbx'] = np.asarray(label.bbx, np.float32).reshape(-1)
        return tfrecord.make_example(feats)

    def shard_name(self, idx: int, num: int) -> str:
        return '%s-%d-of-%d' % (self.subset, idx, num)

    def write_TFRecord(self, num_shards: int, num_threads: int = 1) -> List[str]:
        """``write_TFRecord_multi_thread``: annotations split evenly (np.linspace) over threads, then over each thread's
        shards -- the same shard boundaries as the reference for the same (num_threads, num_shards)."""
        assert num_shards % num_threads == 0, 'please make the num_threads commensurate with file_shards'
        if not self._annotations:
            self.loadAnnotation()
        os.makedirs(self.tf_dir, exist_ok=True)
        per = num_shards // num_threads
        outer = np.linspace(0, len(self._annotations), num_threads + 1).astype(int)
        paths = []
        for t in range(num_threads):
            inner = np.linspace(outer[t], outer[t + 1], per + 1).astype(int)
            for s in range(per):
                path = os.path.join(self.tf_dir, self.shard_name(t * per + s, num_shards))
                tfrecord.write_records(path, (self.convert_to_example(self._annotations[i]) for i in range(inner[s], inner[s + 1])))
                paths.append(path)
        return paths

    # -- reading (dataset_base.py:129-240) -------------------------------------------------------------
    @property
    def filenames(self) -> List[str]:
        pattern = os.path.join(self.tf_dir, '%s-*' % ('testing' if self.subset == 'testing' else 'training'))
        return sorted(glob.glob(pattern))

    @property
    def is_train(self) -> bool:
        return self.subset != 'testing'

    @property
    def approximate_num(self) -> int:
        return self.approximate_num_per_file * len(self.filenames)

    def parse_example(self, example_serialized: bytes):
        """-> (PngInfo, samples uint8 [H][row_bytes], pose float32, name, bbx or None)."""
        f = tfrecord.parse_example(example_serialized)
        pose = np.asarray(f['xyz_pose'], np.float32)
        if pose.size != (self.orig_pose_dim or self.pose_dim):
            raise ValueError('%s: xyz_pose has %d values, expected %d' % (self.name, pose.size, self.orig_pose_dim or self.pose_dim))
        if self.keep_pose_idx is not None:
            pose = pose[self.keep_pose_idx]
        info, samples = png.decode_png(f['png16'][0])
        if (info.height, info.width, info.channels, info.bit_depth) != (self.cfg.h, self.cfg.w, self.png_channels, self.png_depth):
            raise ValueError('%s: frame %dx%d c%d d%d does not match the camera' % (self.name, info.width, info.height, info.channels, info.bit_depth))
        bbx = np.asarray(f['bbx'], np.float32) if 'bbx' in f else None
        return info, samples, pose, f['name'][0].decode(), bbx

    def records(self, shuffle: bool, seed: int = 0, rank: int = 0, world: int = 1, epochs: Optional[int] = None,
                files: Optional[Sequence[str]] = None) -> Iterator[bytes]:
        files = list(files if files is not None else (self.files_override if self.files_override is not None else self.filenames))
        if world > 1:
            files = files[rank::world]
        if not files:
            raise FileNotFoundError('%s: no TFRecord shards under %s' % (self.name, self.tf_dir))
        rng = np.random.default_rng(seed + rank)
        epoch = 0
        while epochs is None or epoch < epochs:
            order = rng.permutation(len(files)) if shuffle else range(len(files))
            buf: List[bytes] = []
            for i in order:
                for rec in tfrecord.read_records(files[i]):
                    if not shuffle:
                        yield rec
                        continue
                    buf.append(rec)
                    if len(buf) >= self.approximate_num_per_file * 8:          # RandomShuffleQueue(min_after_dequeue)
                        j = int(rng.integers(len(buf)))
                        buf[j], buf[-1] = buf[-1], buf[j]
                        yield buf.pop()
            while buf:
                j = int(rng.integers(len(buf)))
                buf[j], buf[-1] = buf[-1], buf[j]
                yield buf.pop()
            epoch += 1

    def preprocess(self, frames, poses, cfgs, bbxs, out_hw: int):
        """``preprocess_op`` (icvl.py / nyu.py:208-221 / msra.py:198-203) on device tensors."""
        from . import preprocess
        return preprocess.crop_and_com_from_pose(frames, poses, cfgs, out_hw, out_hw, dataset=self.name)

    def host_batches(self, batch_size: int, shuffle: bool, seed: int = 0, rank: int = 0, world: int = 1,
                     epochs: Optional[int] = None, drop_last: bool = False, files: Optional[Sequence[str]] = None,
                     workers: int = 4, prefetch: int = 2) -> Iterator[list]:
        """The host half of the pipeline: lists of ``parse_example`` results, one list per batch, in stream order.

        ``workers`` threads decode the frames of a batch concurrently (zlib and ``dr_png_unfilter`` release the GIL; one
        thread decodes ~2600 ICVL or ~500 NYU frames/s, the training step consumes ~1800 crops/s:
        ``profiles/r01_dataio_bench.md``) -- the reference's ``num_preprocess_threads``.  With ``prefetch`` > 0 a
        producer thread stays that many decoded batches ahead of the consumer, so decoding overlaps the GPU step the way
        the reference's queue runners did (``dataset_base.py:153-205``); an exception in the producer is re-raised at the
        consumer's next ``next()``."""
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor

        def produce() -> Iterator[list]:
            pool = ThreadPoolExecutor(workers) if workers > 1 else None
            try:
                pend: List[bytes] = []
                for rec in self.records(shuffle, seed, rank, world, epochs, files):
                    pend.append(rec)
                    if len(pend) == batch_size:
                        yield list(pool.map(self.parse_example, pend)) if pool else [self.parse_example(r) for r in pend]
                        pend = []
                if pend and not drop_last:
                    yield list(pool.map(self.parse_example, pend)) if pool else [self.parse_example(r) for r in pend]
            finally:
                if pool:
                    pool.shutdown(wait=False)

        if prefetch <= 0:
            yield from produce()
            return
        q: 'queue.Queue' = queue.Queue(maxsize=prefetch)
        stop = threading.Event()
        END = object()

        def run():
            try:
                for items in produce():
                    while not stop.is_set():
                        try:
                            q.put(items, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(END)
            except BaseException as e:                                        # handed to the consumer
                q.put(e)

        th = threading.Thread(target=run, name='%s-decode' % self.name, daemon=True)
        th.start()
        try:
            while True:
                items = q.get()
                if items is END:
                    return
                if isinstance(items, BaseException):
                    raise items
                yield items
        finally:                                                              # consumer stopped early: let the producer exit
            stop.set()

    def batches(self, batch_size: int, device, out_hw: int = 128, shuffle: Optional[bool] = None, seed: int = 0, rank: int = 0,
                world: int = 1, epochs: Optional[int] = None, drop_last: bool = False, files: Optional[Sequence[str]] = None,
                workers: int = 4, prefetch: int = 2):
        """``host_batches`` + the device half: the samples go up as bytes, ``dr_depth_from_samples`` and the crop kernel
        produce ``[dm (B,out,out,1), pose (B,3J) (host), cfg (B,6), com (B,3), names]``."""
        import torch
        shuffle = self.is_train if shuffle is None else shuffle
        cfg_row = np.asarray(self.cfg, np.float32)
        for items in self.host_batches(batch_size, shuffle, seed, rank, world, epochs, drop_last, files, workers, prefetch):
            info = items[0][0]
            samples = torch.from_numpy(np.stack([it[1] for it in items])).to(device)     # bytes go up, not floats
            frames = png.depth_from_samples(samples, info)
            poses = np.stack([it[2] for it in items]).astype(np.float32)
            d_pose = torch.from_numpy(poses).to(device)
            d_cfg = torch.from_numpy(np.tile(cfg_row, (len(items), 1))).to(device)
            bbxs = None if items[0][4] is None else torch.from_numpy(np.stack([it[4] for it in items])).to(device)
            crops, _, new_cfgs, coms = self.preprocess(frames, d_pose, d_cfg, bbxs, out_hw)
            yield crops.unsqueeze(-1), poses, new_cfgs, coms, [it[3] for it in items]

    def batch(self, batch_size: int, index: int, device=None):
        """The drivers' interface (``SyntheticDataset.batch``): the next batch of an endless (training) or single-pass,
        last-batch-padded (testing) stream; ``index`` is ignored -- the stream carries the position."""
        import torch
        if self._iter is None:
            device = device or torch.device('cuda', torch.cuda.current_device())
            self._iter = self.batches(batch_size, device, shuffle=self.is_train, seed=self.seed, rank=self.rank,
                                      world=self.world, epochs=None if self.is_train else 1)
        try:
            return next(self._iter)
        except StopIteration:
            self._iter = None
            raise


class IcvlDataset(BaseDataset):
    """data/icvl.py."""
    name = 'icvl'
    cfg = CameraConfig(fx=241.42, fy=241.42, cx=160, cy=120, w=320, h=240)
    approximate_num_per_file = 220
    max_depth = 500.0
    pose_dim, jnt_num = 48, 16
    directory = './exp/data/icvl/'

    def __init__(self, subset, directory=None):
        if subset not in ('training', 'training_small', 'validation', 'testing'):
            raise ValueError('unknown sub %s set to ICVL hand datset' % subset)
        super().__init__(subset, directory)
        self.src_dir = os.path.join(self.directory, 'Testing' if subset == 'testing' else 'Training')
        self.img_dir = os.path.join(self.src_dir, 'Depth')
        self.tf_dir = os.path.join(self.directory, 'tf_test' if subset == 'testing' else 'tf_train')

    @property
    def filenames(self):
        tr = [os.path.join(self.tf_dir, 'training-%d-of-100' % i) for i in range(100)]
        if self.subset == 'training':
            return tr + [tr[-1]]
        if self.subset == 'training_small':
            return [f for i, f in enumerate(tr[:10]) if i % 10 == 0]
        if self.subset == 'validation':
            return [f for i, f in enumerate(tr[:10]) if i % 21 == 0]
        te = [os.path.join(self.tf_dir, 'testing-%d-of-4' % i) for i in range(4)]
        return te + [te[-1]]

    @property
    def exact_num(self):
        return 1596 if self.subset == 'testing' else self.approximate_num

    def loadAnnotation(self, path: Optional[str] = None):
        """labels.txt: ``name u0 v0 d0 u1 ...`` (icvl.py:96-121): uvd -> xyz with the camera."""
        path = path or os.path.join(self.src_dir, 'labels.txt')
        self._annotations = []
        with open(path) as f:
            for line in f:
                if self.subset != 'testing' and not line.startswith('2014'):
                    continue
                buf = line.split()
                if not buf:
                    continue
                pose = uvd2xyz(np.array([float(d) for d in buf[1:]]), self.cfg).reshape(-1)
                self._annotations.append(Annotation(buf[0], pose.tolist()))
        return self._annotations


class NyuDataset(BaseDataset):
    """data/nyu.py."""
    name = 'nyu'
    cfg = CameraConfig(fx=588.235, fy=587.084, cx=320, cy=240, w=640, h=480)
    approximate_num_per_file = 730
    max_depth = 1500.0
    png_channels, png_depth = 3, 8
    orig_pose_dim = 108
    directory = './exp/data/nyu/'
    _keep = [0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32]

    def __init__(self, subset, directory=None, bbx_path: Optional[str] = None):
        if subset not in ('training', 'training_small', 'validation', 'testing'):
            raise ValueError('unknown sub %s set to NYU hand datset' % subset)
        super().__init__(subset, directory)
        self.src_dir = self.img_dir = os.path.join(self.directory, 'dataset/test' if subset == 'testing' else 'dataset/train')
        self.tf_dir = os.path.join(self.directory, 'tf_test' if subset == 'testing' else 'tf_train')
        self.keep_pose_idx = np.array([3 * j + k for j in self._keep for k in range(3)])
        self.pose_dim = len(self.keep_pose_idx)
        self.jnt_num = self.pose_dim // 3
        self.bbx_path = bbx_path or 'data/nyu_bbx.pkl'

    @property
    def filenames(self):
        tr = [os.path.join(self.tf_dir, 'training-%d-of-300' % i) for i in range(100)]
        if self.subset == 'training':
            return tr + [tr[-1]]
        if self.subset == 'training_small':
            return [f for i, f in enumerate(tr[:30]) if i % 10 == 0]
        if self.subset == 'validation':
            return [f for i, f in enumerate(tr) if i % 21 == 0]
        te = [os.path.join(self.tf_dir, 'testing-%d-of-16' % i) for i in range(16)]
        return te + [te[-1]]

    @property
    def exact_num(self):
        return 8252 if self.subset == 'testing' else self.approximate_num

    def loadAnnotation(self, is_trun: bool = False):
        """joint_data.mat['joint_xyz'] (camera, frame, 36, 3) with y flipped; test boxes from nyu_bbx.pkl (nyu.py:93-139)."""
        import scipy.io as sio
        mat = sio.loadmat(os.path.join(self.src_dir, 'joint_data.mat'))
        cams = 1 if self.subset == 'testing' else 3
        bbxes = None
        if self.subset == 'testing':
            with open(self.bbx_path, 'rb') as f:
                bbxes = np.asarray(pickle.load(f, encoding='latin1'), np.float32).reshape(-1, 5)
        self._annotations = []
        for cam in range(cams):
            joints = np.array(mat['joint_xyz'][cam], np.float64)
            for idx, j in enumerate(joints):
                j = j.reshape(-1, 3).copy()
                j[:, 1] *= -1.0
                j = j.reshape(-1)
                if is_trun:
                    j = j[self.keep_pose_idx]
                name = 'depth_{}_{:07d}.png'.format(cam + 1, idx + 1)
                self._annotations.append(Annotation(name, j, bbxes[idx] if bbxes is not None else None))
        return self._annotations

    def preprocess(self, frames, poses, cfgs, bbxs, out_hw):
        from . import preprocess
        if bbxs is not None:                                                     # testing: nyu.py:209-214
            return preprocess.crop_and_com_from_bbx(frames, poses, bbxs, cfgs, out_hw, out_hw)
        return preprocess.crop_and_com_from_pose(frames, poses, cfgs, out_hw, out_hw, dataset=self.name)


class MsraDataset(BaseDataset):
    """data/msra.py (leave-one-subject-out: ``pid`` is the held-out person)."""
    cfg = CameraConfig(fx=241.42, fy=241.42, cx=160, cy=120, w=320, h=240)
    approximate_num_per_file = 85
    max_depth = 1000.0
    pose_dim, jnt_num = 63, 21
    pose_list = '1 2 3 4 5 6 7 8 9 I IP L MP RP T TIP Y'.split()
    pid_num = [8499, 8492, 8412, 8488, 8500, 8497, 8497, 8498, 8492]
    directory = './exp/data/msra15/'

    def __init__(self, subset, pid, directory=None):
        if subset not in ('training', 'testing'):
            raise ValueError('unknown sub %s set to MSRA hand datset' % subset)
        super().__init__(subset, directory)
        self.pid = pid
        self.name = 'msra'
        self.desc = 'msra_P%d' % pid
        self.src_dir = self.img_dir = os.path.join(self.directory, 'P%d' % pid)
        self.tf_dir = os.path.join(self.directory, 'tf')

    def shard_name(self, idx, num):
        return 'P%d-%d-of-%d' % (self.pid, idx, num)

    @property
    def filenames(self):
        if self.subset == 'training':                                            # every subject but pid (msra.py:47-58)
            files = [os.path.join(self.tf_dir, 'P%d-%d-of-100' % (p, i)) for p in range(9) if p != self.pid for i in range(100)]
            return files + [files[-1]]
        files = [os.path.join(self.tf_dir, 'P%d-%d-of-100' % (self.pid, i)) for i in range(100)]
        return files + [files[-1]]

    @property
    def exact_num(self):
        return self.pid_num[self.pid] if self.subset == 'testing' else self.approximate_num

    def image_bytes(self, label):
        with open(os.path.join(self.img_dir, label.name + '.png'), 'rb') as f:
            return f.read()

    def loadAnnotation(self):
        """P<pid>/<gesture>/joint.txt: first line = frame count, then 63 values per frame, y and z negated (msra.py:86-110)."""
        self._annotations = []
        for pose_name in self.pose_list:
            path = os.path.join(self.src_dir, pose_name, 'joint.txt')
            if not os.path.exists(path):
                continue
            with open(path) as f:
                for frm, line in enumerate(f):
                    if frm == 0:
                        continue
                    v = np.array([float(d) for d in line.split()])
                    v[1::3] *= -1.0
                    v[2::3] *= -1.0
                    self._annotations.append(Annotation(os.path.join(pose_name, '%06i_depth' % (frm - 1)), v.tolist()))
        return self._annotations

    def cvtBin2Png(self):
        """.bin -> 16-bit PNG next to it (msra.py:120-149)."""
        prev = None
        for anno in self._annotations or self.loadAnnotation():
            dm = read_msra_bin(os.path.join(self.img_dir, anno.name + '.bin'), prev)
            prev = dm.copy()
            with open(os.path.join(self.img_dir, anno.name + '.png'), 'wb') as f:
                f.write(png.encode_png(dm.astype(np.uint16), filter_type=2))


def get_dataset(name: str, subset: str, directory: Optional[str] = None, pid: int = 0):
    if name == 'icvl':
        return IcvlDataset(subset, directory)
    if name == 'nyu':
        return NyuDataset(subset, directory)
    if name == 'msra':
        return MsraDataset(subset, pid, directory)
    raise ValueError('unknown dataset %s' % name)
