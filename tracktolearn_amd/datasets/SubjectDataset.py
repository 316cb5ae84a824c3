"""Subject dataset for training.

The reference stores subjects in HDF5 (TrackToLearn/datasets/SubjectDataset.py,
create_dataset.py:40-57): ``/{split}/{subject}/{input,peaks,tracking,seeding,
anat}_volume/data`` with attribute ``vox2rasmm``.  h5py is absent from this
image, so the same tree is also accepted flattened into an ``.npz`` archive:
key ``{split}/{subject}/{volume}/data`` holds the array and
``{split}/{subject}/{volume}/vox2rasmm`` the 4x4 affine
(``write_npz_dataset`` creates one).  ``.hdf5`` / ``.h5`` files are read with
h5py when it is importable.
"""
import numpy as np

from tracktolearn_amd.datasets.utils import MRIDataVolume

_VOLUMES = ('input_volume', 'peaks_volume', 'tracking_volume',
            'seeding_volume', 'anat_volume')


def write_npz_dataset(path, subjects):
    """``subjects``: {split: {subject_id: {volume_name: (data, affine)}}}."""
    flat = {}
    for split, subs in subjects.items():
        for sid, vols in subs.items():
            for name, (data, affine) in vols.items():
                flat[f'{split}/{sid}/{name}/data'] = np.asarray(data)
                flat[f'{split}/{sid}/{name}/vox2rasmm'] = np.asarray(affine)
    np.savez(path, **flat)


class _NpzArchive(object):
    def __init__(self, path, split):
        self.z = np.load(path)
        self.split = split
        self.subjects = sorted({k.split('/')[1] for k in self.z.files
                                if k.startswith(split + '/')})

    def volume(self, subject, name, default=None):
        key = f'{self.split}/{subject}/{name}'
        if key + '/data' not in self.z.files:
            if default is None:
                raise KeyError(key)
            print('Missing {} from dataset'.format(name))
            ref = f'{self.split}/{subject}/{default}'
            return (np.zeros_like(self.z[ref + '/data'], dtype=np.float32),
                    np.array(self.z[ref + '/vox2rasmm'], dtype=np.float32))
        return (np.array(self.z[key + '/data'], dtype=np.float32),
                np.array(self.z[key + '/vox2rasmm'], dtype=np.float32))


class _Hdf5Archive(object):
    def __init__(self, path, split):
        try:
            import h5py
        except ImportError as exc:      # pragma: no cover
            raise ImportError(
                'reading HDF5 datasets needs h5py; convert the dataset to the '
                '.npz layout described in tracktolearn_amd/datasets/'
                'SubjectDataset.py') from exc
        self.f = h5py.File(path, 'r')[split]
        self.split = split
        self.subjects = list(self.f.keys())

    def volume(self, subject, name, default=None):
        grp = self.f[subject]
        if name not in grp:
            if default is None:
                raise KeyError(name)
            print('Missing {} from dataset'.format(name))
            return (np.zeros_like(grp[default]['data'], dtype=np.float32),
                    np.array(grp[default].attrs['vox2rasmm'], dtype=np.float32))
        return (np.array(grp[name]['data'], dtype=np.float32),
                np.array(grp[name].attrs['vox2rasmm'], dtype=np.float32))


class SubjectDataset(object):
    """Random-access subjects of one split; items are the 6-tuple
    (subject_id, input_volume, tracking_mask, seeding, peaks, reference) of
    SubjectDataset.py:31-56.  Volumes are float32 with float32 affines, as
    ``MRIDataVolume.from_hdf_group`` makes them (datasets/utils.py:23-34)."""

    def __init__(self, file_path, dataset_split):
        self.file_path = file_path
        self.split = dataset_split
        lower = str(file_path).lower()
        if lower.endswith('.npz'):
            self.archive = _NpzArchive(file_path, dataset_split)
        else:
            self.archive = _Hdf5Archive(file_path, dataset_split)
        self.subjects = self.archive.subjects

    def __len__(self):
        return len(self.subjects)

    def __getitem__(self, index):
        sid = self.subjects[index]
        a = self.archive
        input_volume = MRIDataVolume(*a.volume(sid, 'input_volume'))
        input_volume.subject_id = sid
        peaks = MRIDataVolume(*a.volume(sid, 'peaks_volume'))
        tracking = MRIDataVolume(*a.volume(sid, 'tracking_volume'))
        seeding = MRIDataVolume(*a.volume(sid, 'seeding_volume', 'tracking_volume'))
        anat = MRIDataVolume(*a.volume(sid, 'anat_volume', 'tracking_volume'))
        reference = {'affine': anat.affine_vox2rasmm, 'shape': anat.shape[:3],
                     'zooms': tuple(np.sqrt((np.asarray(
                         anat.affine_vox2rasmm)[:3, :3] ** 2).sum(0)))}
        return (sid, input_volume, tracking, seeding, peaks, reference)
