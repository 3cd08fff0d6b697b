"""The long GPU cases (-m gpu_long): the same test bodies as tests/test_gpu_fullsize.py at the sizes / round counts that cost tens of
seconds of CPU restatement each. NOT part of `-m gpu` (the driver's metered run), run by hand at least once per round:
    python -m pytest tests -m gpu_long -q
(round 6: profiles/r06_*_pytest_gpu_long.log)."""
import pytest

from tests import test_gpu_fullsize as T
from tests import test_gpu_msm as M

pytestmark = pytest.mark.gpu_long


@pytest.mark.parametrize("curve,logn", T.G2_FULL_RANGE_LONG)
def test_msm_g2_full_range_points_long(gpu, curve, logn):
    T.test_msm_g2_full_range_points_equals_cpu_restatement(gpu, curve, logn)


@pytest.mark.parametrize("logn,ncomp", T.NTT_BEYOND_LONG)
def test_ntt_beyond_2p23_long(gpu, logn, ncomp):
    T.test_ntt_beyond_2p23_equals_cpu_restatement(gpu, logn, ncomp)


@pytest.mark.parametrize("curve,group,family", T.RANDOM_POINTS_LONG)
def test_msm_2p20_random_points_long(gpu, curve, group, family):
    T.test_msm_2p20_random_points_equals_cpu_restatement(gpu, curve, group, family)


@pytest.mark.parametrize("curve,group,rounds", T.FUZZ_LONG)
def test_msm_fuzz_long(gpu, curve, group, rounds):
    T.test_msm_fuzz_sizes_and_plans_vs_cpu_restatement(gpu, curve, group, rounds)


@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
@pytest.mark.parametrize("logn,ncomp", T.NTT_FULL_LONG)
def test_ntt_full_size_long(gpu, curve, logn, ncomp):
    T.test_ntt_full_size_equals_cpu_restatement(gpu, curve, logn, ncomp)


@pytest.mark.parametrize("variant", T.NTT_VARIANTS_LONG)
@pytest.mark.parametrize("curve", ["bn254", "bls12_381"])
def test_ntt_every_size_long(gpu, curve, variant):
    T.test_ntt_every_size_up_to_2p19_vs_cpu_restatement(gpu, curve, variant)


@pytest.mark.parametrize("curve,group", M.GROUPS)
def test_msm_fixed_base_table_layouts_long(gpu, curve, group):
    M.test_msm_fixed_base_tables(gpu, curve, group, M.TABLE_LAYOUTS_LONG)


@pytest.mark.parametrize("curve", T.SHARE_VECTOR_CURVES_LONG)
def test_share_vector_kernels_beyond_one_launch_width_long(gpu, curve):
    T.test_share_vector_kernels_beyond_one_launch_width(gpu, curve)
