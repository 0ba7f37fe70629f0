"""The saved-activation offload with its SHIPPED defaults (one copy stream for both directions, the host thread does not wait at the end of
the forward: copies still in flight are handed back from the device) - the bit-identity check of tests/test_host_offload_gpu.py.  These
settings were chosen in the last device call of round 6 (HO13, bench runs); the file sorts last because this test itself could not be run on a
device before the round closed."""
import pytest

from test_host_offload_gpu import DEFAULT_CASES, check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("free,keep,soft,park,batch,defaults", DEFAULT_CASES)
def test_offload_defaults_give_the_same_bits(free, keep, soft, park, batch, defaults):
    check(free, keep, soft, park, batch, defaults)
