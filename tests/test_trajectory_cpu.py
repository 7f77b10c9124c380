"""CPU: the recorded multi-step trajectory of the real reference's EuclideanGPBandit
(tests/golden/trajectory_branin.npz) replayed through the mirrors over the stand-in engine
(tests/oracle_engine.py; test infrastructure) -- host logic, option handling and random-number
call order of the S2/S3/S4 seams.  The same replay runs on the MI355X in
tests/test_gpu_trajectory.py."""
from trajectory_replay import replay


def test_trajectory_replay_over_the_stand_in_engine(monkeypatch):
  from oracle_engine import patch_engine
  patch_engine(monkeypatch)
  n_fit, n_acq = replay()
  assert n_fit == 4 and n_acq == 14
