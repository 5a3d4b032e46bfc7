"""pytest plugin (PYTHONPATH=tools pytest -p pytest_failnow): prints a failing test's report the moment it fails, to the real
stderr -- for runs whose process may not live to see pytest's own summary (tools/gpu_efence.c sessions)."""
import os


def pytest_runtest_logreport(report):
    if report.failed:
        os.write(2, ("\n==== FAILED NOW: %s (%s)\n%s\n====\n" % (report.nodeid, report.when, report.longreprtext[-6000:])).encode())
