"""Host-side helpers around the Kaldi-style data interface (scp files, per-utterance npz / wav): cheap frame
counts for length-balanced sharding, without decoding the payloads."""
import wave
import zipfile

from numpy.lib import format as npformat


def npz_frames(path, key="mix"):
    """Frame count T of the (257, T) array `key` in a feats npz (steps/extract_feats.py:90 writes zlib-compressed
    npz): only the .npy header of the zip member is inflated."""
    with zipfile.ZipFile(path) as z:
        with z.open(key + ".npy") as f:
            version = npformat.read_magic(f)
            if version == (1, 0):
                shape, _, _ = npformat.read_array_header_1_0(f)
            else:
                shape, _, _ = npformat.read_array_header_2_0(f)
    return int(shape[1]) if len(shape) > 1 else int(shape[0])


def wav_frames(path, hop=128):
    """STFT frame count 1 + N // hop of a wav file, from its header."""
    with wave.open(path, "rb") as w:
        return 1 + w.getnframes() // hop
