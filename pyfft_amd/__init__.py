"""pyfft_amd: MI355X-native batched power-of-two c2c FFT behind pyfft's Plan()/execute() API.

    from pyfft_amd.hip import Plan

`VERSION` mirrors pyfft.VERSION (pyfft/__init__.py:1): a tuple of ints.
"""

VERSION = (0, 1, 0)
