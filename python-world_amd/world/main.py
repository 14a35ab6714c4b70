"""`World` facade — same class, method names, arguments and result-dict keys as world/main.py:26-214
of the reference; every stage it calls runs on the MI355X through libworld_hip.so.

Only the analysis/synthesis path is provided: the reference's feature helpers
(mel filterbank, MCEP, VAE glue, draw) are downstream of encode/decode and are not part of this build.
"""
import logging
from typing import Iterable

import numpy as np

from .cheaptrick import cheaptrick
from .d4c import d4c
from .d4cRequiem import d4cRequiem
from .dio import dio
from .get_seeds_signals import get_seeds_signals
from .harvest import harvest
from .stonemask import stonemask
from .swipe import swipe
from .synthesis import synthesis
from .synthesisRequiem import synthesisRequiem


class World(object):
    def get_f0(self, fs: int, x: np.ndarray, f0_method: str = 'harvest', f0_floor: int = 71, f0_ceil: int = 800,
               channels_in_octave: int = 2, target_fs: int = 4000, frame_period: int = 5) -> tuple:
        """world/main.py:27-49."""
        source = self._f0(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period, None)
        return source['temporal_positions'], source['f0'], source['vuv']

    def get_spectrum(self, fs: int, x: np.ndarray, f0_method: str = 'harvest', f0_floor: int = 71,
                     f0_ceil: int = 800, channels_in_octave: int = 2, target_fs: int = 4000, frame_period: int = 5,
                     fft_size=None) -> dict:
        """world/main.py:51-79."""
        source = self._f0(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period, None)
        filt = cheaptrick(x, fs, source, fft_size=fft_size)
        return {'f0': source['f0'],
                'temporal_positions': source['temporal_positions'],
                'fs': fs,
                'ps spectrogram': filt['ps spectrogram'],
                'spectrogram': filt['spectrogram']}

    def encode_w_gvn_f0(self, fs: int, x: np.ndarray, source: dict, fft_size=None, is_requiem: bool = False) -> dict:
        """world/main.py:81-104 (including its quirks: fft_size=None raises TypeError at the assert and
        is_requiem=True raises KeyError('coarse_ap'))."""
        assert np.all(source['f0'] >= 3 * fs / fft_size)
        filt = cheaptrick(x, fs, source, fft_size=fft_size)
        if is_requiem:
            source = d4cRequiem(x, fs, source, fft_size=fft_size)
        else:
            source = d4c(x, fs, source, fft_size_for_spectrum=fft_size)
        return {'temporal_positions': source['temporal_positions'],
                'vuv': source['vuv'],
                'f0': source['f0'],
                'fs': fs,
                'spectrogram': filt['spectrogram'],
                'aperiodicity': source['aperiodicity'],
                'coarse_ap': source['coarse_ap'],
                'is_requiem': is_requiem}

    def encode(self, fs: int, x: np.ndarray, f0_method: str = 'harvest', f0_floor: int = 71, f0_ceil: int = 800,
               channels_in_octave: int = 2, target_fs: int = 4000, frame_period: int = 5,
               allowed_range: float = 0.1, fft_size=None, is_requiem: bool = False) -> dict:
        """world/main.py:106-152."""
        if fft_size != None:  # noqa: E711  (same test as the reference)
            f0_floor = 3.0 * fs / fft_size
        source = self._f0(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period,
                          allowed_range)
        filt = cheaptrick(x, fs, source, fft_size=fft_size)
        if is_requiem:
            source = d4cRequiem(x, fs, source, fft_size=fft_size)
        else:
            source = d4c(x, fs, source, fft_size_for_spectrum=fft_size)
        return {'temporal_positions': source['temporal_positions'],
                'vuv': source['vuv'],
                'fs': filt['fs'],
                'f0': source['f0'],
                'aperiodicity': source['aperiodicity'],
                'ps spectrogram': filt['ps spectrogram'],
                'spectrogram': filt['spectrogram'],
                'is_requiem': is_requiem}

    @staticmethod
    def _f0(fs, x, f0_method, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period, allowed_range):
        if f0_method == 'dio':
            if allowed_range is None:
                source = dio(x, fs, f0_floor, f0_ceil, channels_in_octave, target_fs, frame_period)
            else:
                source = dio(x, fs, f0_floor=f0_floor, f0_ceil=f0_ceil, channels_in_octave=channels_in_octave,
                             target_fs=target_fs, frame_period=frame_period, allowed_range=allowed_range)
            source['f0'] = stonemask(x, fs, source['temporal_positions'], source['f0'])
        elif f0_method == 'harvest':
            source = harvest(x, fs, f0_floor=f0_floor, f0_ceil=f0_ceil, frame_period=frame_period)
        elif f0_method == 'swipe':
            source = swipe(fs, x, plim=[f0_floor, f0_ceil], sTHR=0.3)
        else:
            raise Exception
        return source

    def scale_pitch(self, dat: dict, factor: float) -> dict:
        """In place, returns the same dict (world/main.py:154-162)."""
        dat['f0'] *= factor
        return dat

    def set_pitch(self, dat: dict, time: np.ndarray, value: np.ndarray) -> dict:
        raise NotImplementedError  # world/main.py:164-165

    def scale_duration(self, dat: dict, factor: float) -> dict:
        """In place, returns the same dict (world/main.py:170-178)."""
        dat['temporal_positions'] *= factor
        return dat

    def modify_duration(self, dat: dict, from_time: Iterable, to_time: Iterable) -> dict:
        """world/main.py:180-189 (returns None like the reference)."""
        end = dat['temporal_positions'][-1]
        assert np.all(np.diff(from_time)) > 0
        assert np.all(np.diff(to_time)) > 0
        assert from_time[0] > 0
        assert from_time[-1] < end
        from_time = np.r_[0, from_time, end]
        if to_time[-1] == -1:
            to_time[-1] = end
        dat['temporal_positions'] = np.interp(dat['temporal_positions'], from_time, to_time)

    def warp_spectrum(self, dat: dict, factor: float) -> dict:
        """world/main.py:191-196."""
        k = dat['spectrogram'].shape[0]
        grid = np.arange(0, k) / k
        dat['spectrogram'][:] = np.array([np.interp(grid ** factor, grid, s) for s in dat['spectrogram'].T]).T
        return dat

    def decode(self, dat: dict) -> dict:
        """world/main.py:198-214."""
        if dat['is_requiem']:
            seeds_signals = get_seeds_signals(dat['fs'])
            y = synthesisRequiem(dat, dat, seeds_signals)
        else:
            y = synthesis(dat, dat)
        m = np.max(np.abs(y))
        if m > 1.0:
            logging.info('rescaling waveform')
            y /= m
        dat['out'] = y
        return dat
