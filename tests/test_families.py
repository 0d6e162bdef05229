"""Built-in tag families: layout, uniqueness, rotational Hamming distance, and the upstream anchors (families.c header)."""
import numpy as np

from chalkydri_amd import family


def _codes(name):
    f = family(name).contents
    return f, np.array([f.codes[i] for i in range(f.ncodes)], dtype=np.uint64)


def _rot(codes, nbits):
    q = nbits // 4
    mask = np.uint64((1 << nbits) - 1)
    return ((codes << np.uint64(q)) | (codes >> np.uint64(3 * q))) & mask


def _min_rotational_hamming(codes, nbits):
    rots = [codes]
    for _ in range(3):
        rots.append(_rot(rots[-1], nbits))
    allc = np.concatenate(rots)                       # every code in its four orientations
    best = 99
    pop = np.vectorize(lambda v: bin(int(v)).count("1"))
    for i, c in enumerate(codes):
        d = pop(allc ^ c)
        d[i] = 99                                     # the code itself at rotation 0
        best = min(best, int(d.min()))
    return best


def test_tag16h5_is_upstream_and_distance_5():
    f, codes = _codes("tag16h5")
    assert (f.nbits, f.ncodes, f.width_at_border, f.total_width, f.reversed_border, f.min_hamming) == (16, 30, 6, 8, 0, 5)
    assert len(set(codes.tolist())) == 30
    assert codes[0] == 0x27c8 and codes[1] == 0x31b6 and codes[29] == 0xb57a      # AprilTag-3 tag16h5.c anchors
    assert _min_rotational_hamming(codes, 16) == 5


def test_tag36h11_layout_anchors_and_distance_11():
    f, codes = _codes("tag36h11")
    assert (f.nbits, f.ncodes, f.width_at_border, f.total_width, f.reversed_border, f.min_hamming) == (36, 587, 8, 10, 0, 11)
    assert len(set(codes.tolist())) == 587
    head = [0xd7e00984b, 0xdda664ca7, 0xdc4a1c821, 0xe17b470e9, 0xef91d01b1, 0xf429cdd73, 0x005da29225, 0x1106cba43,
            0x223bed79d, 0x21f51213c, 0x33eb19ca6, 0x3f76eb0f8, 0x469a97414]
    assert codes[:13].tolist() == head                                              # upstream IDs 0..12
    bx = [f.bit_x[i] for i in range(36)]
    by = [f.bit_y[i] for i in range(36)]
    assert sorted(zip(bx, by)) == sorted((x, y) for x in range(1, 7) for y in range(1, 7))   # the 6x6 data area, once each
    # the layout is rotation-symmetric: rotating the tag by 90 degrees rotates the code by nbits/4
    for i in range(36):
        j = (i + 9) % 36
        assert (bx[j], by[j]) == (7 - by[i], bx[i])
    assert _min_rotational_hamming(codes, 36) == 11


def test_unknown_family():
    import pytest
    with pytest.raises(KeyError):
        family("tag25h9")
