"""create_dataset / create_dataloader - mirror of the reference's data/__init__.py:9-53.

Modes: 'Synthetic_RGGB2BGR' (no files needed) and the reference's four RGGB -> BGR datasets
(SID_Sony_Ratio[_Test]_RGGB2BGR, S7ISP_RGGB2BGR[_Test]; data/rggb2bgr_datasets.py - .png / .npy frames read without
cv2, lmdb when the module is present).  The two OnePlus_Rggb2Obj modes feed the YOLOv3 detection loss, which is
outside the hot-path scope, and raise.  The loader contract is the reference's: train loaders
shard ``batch_size`` over the ranks and drop the last batch, test loaders yield one image at a time."""
import logging

import torch
import torch.distributed as dist
import torch.utils.data
from torch.utils.data.dataloader import default_collate

_RGGB2BGR = ('SID_Sony_Ratio_RGGB2BGR', 'SID_Sony_Ratio_Test_RGGB2BGR', 'S7ISP_RGGB2BGR', 'S7ISP_RGGB2BGR_Test')
_DETECTION = ('OnePlus_Rggb2Obj', 'OnePlus_Rggb2Obj_Test')


def create_dataloader(dataset, dataset_opt, opt=None, sampler=None, collate_fn=None):
    phase = dataset_opt['phase']
    if phase == 'test':
        return torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0, pin_memory=True)
    if phase != 'train':
        raise ValueError('Unknown phase: {}'.format(phase))
    if opt['dist']:
        world = dist.get_world_size()
        assert dataset_opt['batch_size'] % world == 0, 'batch_size must divide over the ranks'
        batch, workers, shuffle = dataset_opt['batch_size'] // world, dataset_opt['n_workers'], False
    else:
        batch = dataset_opt['batch_size']
        workers = dataset_opt['n_workers'] * len(opt['gpu_ids'] or [0])
        shuffle = sampler is None
    if dataset_opt.get('device_resident') and hasattr(dataset, '_frames') and torch.cuda.is_available():
        # frames stay in HBM, a batch is two gather kernels (data/device_loader.py); the sampler only says WHICH frames
        from .device_loader import DeviceCropLoader
        frames = list(sampler) if sampler is not None else list(range(len(dataset)))
        return DeviceCropLoader(dataset, batch, frames, torch.device('cuda'))
    return torch.utils.data.DataLoader(dataset, batch_size=batch, shuffle=shuffle, num_workers=workers, sampler=sampler,
                                       drop_last=True, pin_memory=False, collate_fn=collate_fn or default_collate)


def create_dataset(dataset_opt):
    mode = dataset_opt['mode']
    if mode == 'Synthetic_RGGB2BGR':
        from .synthetic_raw import SyntheticRawDataset as D
    elif mode in _RGGB2BGR:
        from .rggb2bgr_datasets import Rggb2BgrDataset as D
    elif mode in _DETECTION:
        raise NotImplementedError(
            'Dataset [{:s}] pairs RAW frames with detection labels for the YOLOv3 task loss, which is outside the scope '
            'of this build; the RGGB2BGR modes and Synthetic_RGGB2BGR are available.'.format(mode))
    else:
        raise NotImplementedError('Dataset [{:s}] is not recognized.'.format(mode))
    dataset = D(dataset_opt)
    logging.getLogger('base').info('Dataset {:s} is created.'.format(mode))
    return dataset
