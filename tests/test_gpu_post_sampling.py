"""MI355X: stand-alone 'post_sampling' hyper-parameter tuning and additive rand_exp_sampling with
every density a device fit (dfh_gp_lml_batch), against the real reference's fitter under the same
seed: the same samples, groupings and random-stream position (tests/post_sampling_replay.py)."""
import pytest

from post_sampling_replay import POST_SAMPLING_CASES, check_additive_rand_exp_sampling, check_post_sampling

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(POST_SAMPLING_CASES))
def test_post_sampling_draws_the_reference_samples(engine, name):
  check_post_sampling(name)


def test_additive_rand_exp_sampling_draws_the_reference_samples(engine):
  check_additive_rand_exp_sampling()
