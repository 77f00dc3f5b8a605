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
    """Run the suite on three pytest-xdist workers by default, `--dist loadgroup`: the two files of multi-process tests (up to eight ranks, each a
    process with its own HIP context on the one GPU) are a group each, i.e. each runs serially on ONE worker -- many rank processes of several tests at
    once oversubscribe the GPU's queues and every test slows down (measured: four free workers gained 15 % on the serial run) -- while the single-process
    tests (kernels, engines against the CPU oracle, entry points) fill the third worker and the gaps.  `-n N` on the command line wins;
    IE_TEST_SERIAL=1 or a missing xdist plugin runs serially."""
    if (config.pluginmanager.hasplugin("xdist") and getattr(config.option, "numprocesses", None) is None and not os.environ.get("PYTEST_XDIST_WORKER")
            and os.environ.get("IE_TEST_SERIAL") != "1" and not getattr(config.option, "collectonly", False)):
        config.option.numprocesses = 3
        config.option.dist = "loadgroup"


# Order of the run: the two groups of multi-process tests first (one worker each, for minutes), then the single-process tests that hold the GPU / the
# host cores longest (7B-width shapes against the CPU oracle), then everything else in file order.
_GROUPS = {"test_multirank_gpu.py": "ranks_a", "test_dp_gpu.py": "ranks_b", "test_internlm1_gpu.py": "ranks_b"}
_LONGEST_FIRST = ("test_engine_7b_width_merged_benchmark_step_matches_oracle", "test_engine_7b_shaped_layer_full_size_matches_oracle",
                  "test_first_steps_of_the_benchmark_recipe_retrace_the_oracle_at_7b_width", "test_flash_attention_benchmark_regime_matches_oracle")


def pytest_collection_modifyitems(config, items):
    rank = {n: i for i, n in enumerate(_LONGEST_FIRST)}

    def key(it):
        grp = _GROUPS.get(os.path.basename(str(it.fspath)))
        if grp is not None and it.get_closest_marker("gpu") is not None:
            it.add_marker(pytest.mark.xdist_group(grp))
            return (0, grp, 0)
        return (1, "", rank.get(getattr(it, "originalname", None) or it.name, len(rank)))

    items.sort(key=key)   # (stable: inside a group and among the rest the file order stays)


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
