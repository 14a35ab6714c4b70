"""MI355X-native WORLD vocoder: drop-in mirror of tuanad121/Python-WORLD's `world` package.

`from world import main; main.World().encode(fs, x)` keeps working; every stage function runs as
hand-written HIP kernels (libworld_hip.so) on an AMD MI355X.  There is no CPU fallback.
"""
