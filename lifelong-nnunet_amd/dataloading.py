"""Real-data step BEFORE the hot path: nnU-Net-preprocessed cases -> training patches.

What the reference does (call sites in the reference tree; the implementations are upstream nnU-Net v1 @ 77bc485,
``nnunet/training/dataloading/dataset_loading.py`` and ``nnunet/training/network_training/nnUNetTrainer.py``, absent
from ``/root/reference`` and restated here from their published form -- PARITY UNPINNED, pinned only by the property tests
in tests/test_host_logic.py):

  * ``load_dataset(folder_with_preprocessed_data)`` (rehearsal/nnUNetTrainerRehearsal.py:127): one entry per
    ``<case>.npz`` with ``data_file`` / ``properties_file``;
  * ``do_split()`` (REH.py:128): ``splits_final.pkl`` if present, else 5-fold ``KFold(shuffle=True, random_state=12345)``
    over the sorted case identifiers;
  * ``DataLoader3D(dataset, basic_generator_patch_size, patch_size, batch_size, False,
    oversample_foreground_percent=..., pad_mode="constant", pad_sides=..., memmap_mode='r')`` (REH.py:151-156):
    ``batch_size`` cases drawn with replacement; the LAST ``round(batch * oversample)`` samples of a batch are centred on a
    random voxel of a random foreground class (``properties['class_locations']``), the others cropped uniformly; crops
    may leave the volume by ``(basic - final) / 2`` and are padded (image: ``pad_mode``, segmentation: -1);
  * deep-supervision targets: the segmentation resized with order 0 to ``patch / 2^i`` (``DownsampleSegForDSTransform2``;
    ``skimage.transform.resize`` pixel-centre mapping, i.e. the voxel at ``floor(2^i * (o + 0.5))``).

NOT built (a hook takes one: ``PreprocessedDataProvider(train_transform=...)``): the batchgenerators augmentation pipeline (spatial / intensity transforms), the
2-D loader, cascade inputs.  ``PreprocessedDataProvider`` adapts the loader to the trainers' ``data_provider(task, split,
plans)`` contract and yields the dictionary the iteration consumes (MH.py:606-608).

Attribution: the algorithms restated in this file (function names kept so that callers read like upstream's) are those of
nnU-Net v1 (https://github.com/MIC-DKFZ/nnUNet, commit 77bc485, Apache License 2.0, Isensee et al., Nature Methods 2021);
no upstream source text is included.
"""
from __future__ import annotations

import os
import pickle
from collections import OrderedDict
from typing import Dict

import numpy as np


def load_pickle(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def get_case_identifiers(folder):
    return sorted(f[:-4] for f in os.listdir(folder) if f.endswith(".npz") and f.find("segFromPrevStage") == -1)


def load_dataset(folder, num_cases_properties_loading_threshold=1000):
    case_identifiers = get_case_identifiers(folder)
    dataset = OrderedDict()
    for c in case_identifiers:
        dataset[c] = OrderedDict()
        dataset[c]['data_file'] = os.path.join(folder, "%s.npz" % c)
        dataset[c]['properties_file'] = os.path.join(folder, "%s.pkl" % c)
    if len(case_identifiers) <= num_cases_properties_loading_threshold:
        for c in dataset:
            dataset[c]['properties'] = load_pickle(dataset[c]['properties_file'])
    return dataset


def unpack_dataset(folder, key="data"):
    """npz -> npy next to it (memory-mappable), skipping cases already unpacked."""
    for c in get_case_identifiers(folder):
        npz, npy = os.path.join(folder, c + ".npz"), os.path.join(folder, c + ".npy")
        if not os.path.isfile(npy):
            np.save(npy, np.load(npz)[key])


def do_split(dataset: Dict, fold=0, splits_file=None):
    """(dataset_tr, dataset_val) for ``fold`` -- 'all' trains and validates on everything."""
    keys = np.sort(list(dataset.keys()))
    if fold == "all":
        tr_keys = val_keys = list(keys)
    else:
        if splits_file is not None and os.path.isfile(splits_file):
            splits = load_pickle(splits_file)
        else:
            from sklearn.model_selection import KFold
            splits = []
            for train_idx, test_idx in KFold(n_splits=5, shuffle=True, random_state=12345).split(keys):
                splits.append(OrderedDict(train=np.array(keys)[train_idx], val=np.array(keys)[test_idx]))
            if splits_file is not None:
                with open(splits_file, "wb") as f:
                    pickle.dump(splits, f)
        if fold < len(splits):
            tr_keys, val_keys = list(splits[fold]['train']), list(splits[fold]['val'])
        else:       # upstream: random 80:20 split seeded with 12345 + fold
            rnd = np.random.RandomState(seed=12345 + fold)
            idx_tr = rnd.choice(len(keys), int(len(keys) * 0.8), replace=False)
            tr_keys = [keys[i] for i in idx_tr]
            val_keys = [keys[i] for i in range(len(keys)) if i not in idx_tr]
    tr_keys, val_keys = sorted(tr_keys), sorted(val_keys)
    return OrderedDict((k, dataset[k]) for k in tr_keys), OrderedDict((k, dataset[k]) for k in val_keys)


class DataLoader3D:
    def __init__(self, data, patch_size, final_patch_size, batch_size, has_prev_stage=False,
                 oversample_foreground_percent=0.0, memmap_mode="r", pad_mode="edge", pad_kwargs_data=None, pad_sides=None):
        assert not has_prev_stage, "cascade inputs are out of scope"
        self._data = data
        self.batch_size = batch_size
        self.patch_size, self.final_patch_size = np.array(patch_size), np.array(final_patch_size)
        self.oversample_foreground_percent = oversample_foreground_percent
        self.list_of_keys = list(self._data.keys())
        self.need_to_pad = (self.patch_size - self.final_patch_size).astype(int)
        if pad_sides is not None:
            self.need_to_pad = self.need_to_pad + np.array(pad_sides)
        self.memmap_mode = memmap_mode
        self.pad_mode = pad_mode
        self.pad_kwargs_data = pad_kwargs_data or OrderedDict()
        self.num_channels = None

    def get_do_oversample(self, batch_idx):
        return not batch_idx < round(self.batch_size * (1 - self.oversample_foreground_percent))

    def _load_case(self, i):
        entry = self._data[i]
        npy = entry['data_file'][:-4] + ".npy"
        if os.path.isfile(npy):
            return np.load(npy, self.memmap_mode)
        return np.load(entry['data_file'])['data']

    def generate_train_batch(self):
        selected_keys = np.random.choice(self.list_of_keys, self.batch_size, True, None)
        data, seg, case_properties = None, None, []
        ps = self.patch_size
        for j, i in enumerate(selected_keys):
            force_fg = self.get_do_oversample(j)
            properties = self._data[i]['properties'] if 'properties' in self._data[i] else load_pickle(self._data[i]['properties_file'])
            case_properties.append(properties)
            case_all_data = self._load_case(i)
            if data is None:
                self.num_channels = case_all_data.shape[0] - 1
                data = np.zeros((self.batch_size, self.num_channels) + tuple(ps), dtype=np.float32)
                seg = np.zeros((self.batch_size, 1) + tuple(ps), dtype=np.float32)
            need_to_pad = self.need_to_pad.copy()
            for d in range(3):
                if need_to_pad[d] + case_all_data.shape[d + 1] < ps[d]:
                    need_to_pad[d] = ps[d] - case_all_data.shape[d + 1]
            shape = case_all_data.shape[1:]
            lb = [-need_to_pad[d] // 2 for d in range(3)]
            ub = [shape[d] + need_to_pad[d] // 2 + need_to_pad[d] % 2 - ps[d] for d in range(3)]
            voxels_of_that_class = None
            if force_fg:
                if 'class_locations' not in properties.keys():
                    raise RuntimeError("Please rerun the preprocessing with the newest version of nnU-Net!")
                foreground_classes = np.array([c for c in properties['class_locations'].keys()
                                               if len(properties['class_locations'][c]) != 0])
                foreground_classes = foreground_classes[foreground_classes > 0]
                if len(foreground_classes) > 0:
                    selected_class = np.random.choice(foreground_classes)
                    voxels_of_that_class = properties['class_locations'][selected_class]
            if voxels_of_that_class is not None:
                selected_voxel = voxels_of_that_class[np.random.choice(len(voxels_of_that_class))]
                bbox_lb = [max(lb[d], selected_voxel[d] - ps[d] // 2) for d in range(3)]
            else:
                bbox_lb = [np.random.randint(lb[d], ub[d] + 1) for d in range(3)]
            bbox_ub = [bbox_lb[d] + ps[d] for d in range(3)]
            valid_lb = [max(0, bbox_lb[d]) for d in range(3)]
            valid_ub = [min(shape[d], bbox_ub[d]) for d in range(3)]
            crop = np.copy(case_all_data[:, valid_lb[0]:valid_ub[0], valid_lb[1]:valid_ub[1], valid_lb[2]:valid_ub[2]])
            pads = ((0, 0),) + tuple((-min(0, bbox_lb[d]), max(bbox_ub[d] - shape[d], 0)) for d in range(3))
            data[j] = np.pad(crop[:-1], pads, self.pad_mode, **self.pad_kwargs_data)
            seg[j, 0] = np.pad(crop[-1:], pads, 'constant', **{'constant_values': -1})[0]
        return {'data': data, 'seg': seg, 'properties': case_properties, 'keys': selected_keys}

    def __iter__(self):
        return self

    def __next__(self):
        return self.generate_train_batch()


def downsample_seg_for_ds(seg: np.ndarray, num_pool: int, pool_op_kernel_sizes=None):
    """``DownsampleSegForDSTransform2`` with upstream's ``deep_supervision_scales`` (1 / the cumulative pooling strides per axis,
    2^i for the isotropic plans) and order 0: nearest neighbour at the pixel-centre coordinate ``f * (o + 0.5) - 0.5``, rounded
    half up -> input index ``f * o + f / 2`` (clipped)."""
    from .synthetic import ds_strides
    out = [seg]
    for fz, fy, fx in ds_strides(num_pool, pool_op_kernel_sizes)[1:]:
        fs = (fz, fy, fx)
        new_shape = [int(round(s / f)) for s, f in zip(seg.shape[2:], fs)]
        idx = [np.minimum(np.floor(f * (np.arange(n) + 0.5)).astype(int), s - 1) for n, s, f in zip(new_shape, seg.shape[2:], fs)]
        out.append(np.ascontiguousarray(seg[:, :, idx[0]][:, :, :, idx[1]][:, :, :, :, idx[2]]))
    return out


class PreprocessedDataProvider:
    """``data_provider(task, split, plans)`` for the trainers: ``folders[task]`` = that task's preprocessed folder
    (``<preprocessing_output_dir>/<task>/<data_identifier>_stage<k>``).  No augmentation of its own: final patches are cropped
    directly (``basic_generator_patch_size == patch_size``).  ``train_transform`` / ``val_transform`` are the hook at the point where
    the reference wraps its loaders in upstream's augmentation pipeline (MH.py:904-922 -> ``get_moreDA_augmentation``): a
    batchgenerators-style callable ``transform(**batch) -> batch`` over the loader's dictionary (``data`` (B,C,D,H,W) float32, ``seg``
    (B,1,D,H,W) with -1 outside the volume, ``properties``, ``keys``; numpy), applied BEFORE the two steps upstream's pipeline ends
    with, which stay here: label -1 -> 0 and the deep-supervision down-sampling of ``seg``.  A ``batchgenerators`` ``Compose`` of
    spatial / intensity transforms drops in unchanged."""

    def __init__(self, folders: Dict[str, str], fold=0, oversample_foreground_percent=0.33, unpack=True, extra_train: Dict = None,
                 train_transform=None, val_transform=None):
        self.folders, self.fold, self.oversample = dict(folders), fold, oversample_foreground_percent
        self.unpack = unpack
        self.extra_train = extra_train or {}         # task -> dataset entries mixed in (rehearsal, REH.py:130-136)
        self.train_transform, self.val_transform = train_transform, val_transform

    # ---- the three hooks nnUNetTrainerRehearsal uses to build its fused training set from real folders (REH.py:105-164)
    def dataset_for(self, task):
        folder = self.folders[str(task)]
        if self.unpack:
            unpack_dataset(folder)
        return load_dataset(folder)

    def splits_file_for(self, task):
        return os.path.join(os.path.dirname(self.folders[str(task)]), "splits_final.pkl")

    def generator_for(self, dataset, plans, split="train"):
        loader = DataLoader3D(dataset, plans["patch_size"], plans["patch_size"], plans["batch_size"], False,
                              oversample_foreground_percent=self.oversample, pad_mode="constant", memmap_mode='r')
        return _DictAdapter(loader, plans["num_pool"], plans.get("pool_op_kernel_sizes"),
                            self.train_transform if split == "train" else self.val_transform)

    def __call__(self, task, split, plans):
        tr, val = do_split(self.dataset_for(task), self.fold, self.splits_file_for(task))
        ds = OrderedDict(tr) if split == "train" else val
        if split == "train":
            ds.update(self.extra_train.get(str(task), {}))
        return self.generator_for(ds, plans, split)


class _DictAdapter:
    def __init__(self, loader, num_pool, pool_op_kernel_sizes=None, transform=None):
        self.loader, self.num_pool, self.pools, self.transform = loader, num_pool, pool_op_kernel_sizes, transform

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        b = next(self.loader)
        if self.transform is not None:
            b = self.transform(**b)
        seg = np.maximum(b['seg'], 0)               # -1 (outside the volume) trains as background, as upstream's
        targets = downsample_seg_for_ds(seg, self.num_pool, self.pools)     # RemoveLabelTransform(-1, 0) does
        return {'data': torch.from_numpy(b['data']), 'target': [torch.from_numpy(t) for t in targets],
                'keys': list(b['keys']), 'properties': b['properties']}
