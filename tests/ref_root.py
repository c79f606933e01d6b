"""TEST INFRASTRUCTURE: where the reference tree is.

`/root/reference/dm_control` in the build container.  The GPU box has no `/root/reference`; `scripts/stage_reference.sh`
stages the reference's Python sources and XML assets (4 MB: no meshes, no mocap data) in `_refstage/dm_control` -- a
git-ignored scratch directory that is NOT in .gpurunignore, so it travels with the `gpurun` snapshot and is never
committed -- and the reference-driven tests then run there on the HIP path.  `DMC_REFERENCE_ROOT` overrides both."""
import os

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CANDIDATES = (os.environ.get('DMC_REFERENCE_ROOT'), '/root/reference/dm_control', os.path.join(_REPO, '_refstage', 'dm_control'))


def find():
  for c in _CANDIDATES:
    if c and os.path.isdir(c):
      return c
  return '/root/reference/dm_control'


REF = find()
