import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import __graft_entry__ as entry  # noqa: E402

nhp = entry.load_package()


def show(title, truth, estimate):
    print(title)
    print(np.column_stack([truth, estimate])[:12])
