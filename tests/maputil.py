"""Test-side names of the map helpers (they live in the package since round 5 so that bench.py and tools/ do not import from tests/)."""
from lsc_planner_amd.maps import forest_leaves, write_bt  # noqa: F401
