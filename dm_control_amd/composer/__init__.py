"""Composer-side seam for batches on device: the environment loop with the reference's
hook order (`environment.Environment`), `bind()`-style named views over the SoA fields (`physics.DevicePhysics`),
and the task layers of BASELINE configs 4 and 5 (`tasks.go_to_target`, `tasks.soccer`)."""
from dm_control_amd.composer.environment import Entity, Environment, Task, TimeStep, FIRST, MID, LAST  # noqa: F401

_ENVIRONMENTS = ('cmu_go_to_target', 'soccer_2v2')


def make(name, batch_size, **kwargs):
  """Device-resident batched environment of a BASELINE locomotion config."""
  if name == 'cmu_go_to_target':
    from dm_control_amd.composer.tasks import go_to_target
    return go_to_target.make(batch_size, **kwargs)
  if name == 'soccer_2v2':
    from dm_control_amd.composer.tasks import soccer
    return soccer.make(batch_size, **kwargs)
  raise ValueError('unknown environment %r; available: %s' % (name, _ENVIRONMENTS))
