"""The driver's two entry points keep working: build() here (CPU, cross-compile), smoke() on the GPU."""
import pytest


def test_build_entry():
    import __graft_entry__ as g

    g.build()


@pytest.mark.gpu
def test_smoke_entry():
    import __graft_entry__ as g

    g.smoke()
