"""The driver fixture used by tests/test_run_karate_gpu.py is the reference's examples/run_karate.py, byte for byte."""
import os
import pytest
from conftest import golden_path


def test_fixture_is_byte_identical_to_the_reference_example():
    ref = '/root/reference/examples/run_karate.py'
    if not os.path.exists(ref):
        pytest.skip('reference tree not present on this box')
    assert open(ref, 'rb').read() == open(golden_path('ref_examples_run_karate.py.txt'), 'rb').read()
