import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def xport(port):
    """A rendezvous port private to this pytest-xdist worker (spawned ranks inherit PYTEST_XDIST_WORKER): tests that run at the same time are on
    different workers, so their process groups never meet on a port."""
    w = os.environ.get("PYTEST_XDIST_WORKER", "gw0")
    return int(port) + 1000 * int(w[2:] or 0)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """Run the suite on four pytest-xdist workers by default (the GPU suite is host-bound: process spawns, imports, CPU oracles; one MI355X and
    288 GB hold four tests at a time).  `-n N` on the command line wins; IE_TEST_SERIAL=1 or a missing xdist plugin runs serially."""
    if (config.pluginmanager.hasplugin("xdist") and getattr(config.option, "numprocesses", None) is None and not os.environ.get("PYTEST_XDIST_WORKER")
            and os.environ.get("IE_TEST_SERIAL") != "1" and not getattr(config.option, "collectonly", False)):
        config.option.numprocesses = 4
        config.option.dist = "load"


# the tests that hold the GPU / the host cores for minutes (7B-width shapes against the CPU oracle, eight ranks at 32 768 tokens): they go FIRST, one per
# worker, so that the many small multi-process tests fill in around them instead of waiting behind them at the end of the run
_LONGEST_FIRST = ("test_engine_7b_width_merged_benchmark_step_matches_oracle", "test_isp_config3_layout_seq32768_sp8_at_7b_width",
                  "test_engine_7b_shaped_layer_full_size_matches_oracle", "test_sequence_parallel_sp4_sp8_equals_single_rank_step",
                  "test_llama2_tensor_parallel_2_with_hybrid_zero_on_8_ranks", "test_first_steps_of_the_benchmark_recipe_retrace_the_oracle_at_7b_width")


def pytest_collection_modifyitems(config, items):
    rank = {n: i for i, n in enumerate(_LONGEST_FIRST)}
    items.sort(key=lambda it: rank.get(it.originalname or it.name, len(rank)))   # (stable: everything else keeps its order)


@pytest.fixture
def all_host_cores():
    """A test whose CPU oracle works through a 7B-width step: all host cores for it, whatever share the xdist worker was given."""
    import torch

    before = torch.get_num_threads()
    torch.set_num_threads(os.cpu_count() or before)
    yield
    torch.set_num_threads(before)


def pytest_configure(config):
    n = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "1"))
    if n > 1:   # the CPU oracles of concurrent tests share the host's cores
        import torch

        torch.set_num_threads(max(1, (os.cpu_count() or n) // n))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ranks(n): GPUs a multi-rank test needs for its RCCL (nccl backend) variant")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
