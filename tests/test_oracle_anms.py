"""CPU tests of the ANMS part of the oracle (FeatureDetectorANMS + cv::circle + cv::sortIdx +
anms::Ssc restatements).  Parity unpinned (no OpenCV here, the reference holds no vectors): what
pins these is (i) the toolchain's own std::sort for the degenerate sort (Q9 / SURVEY A.5), (ii) an
independent pure-Python transcription of anms.cpp:344-438, (iii) structural properties."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_circle_shape_and_halfwidths(oracle):
    hw = oracle.circle_halfwidths(10)
    # midpoint walk of cv::Circle for radius 10 (single-pixel tips are OpenCV's known look)
    assert hw.tolist() == [10, 9, 9, 9, 9, 8, 8, 7, 6, 4, 0]
    img = np.ones((41, 41), np.uint8)
    oracle.circle_filled(img, 20, 20, 10)
    zero = img == 0
    assert (zero == zero[::-1]).all() and (zero == zero[:, ::-1]).all() and (zero == zero.T).all()
    for dy in range(-10, 11):
        row = np.flatnonzero(zero[20 + dy])
        assert row.min() == 20 - hw[abs(dy)] and row.max() == 20 + hw[abs(dy)]
        assert len(row) == 2 * hw[abs(dy)] + 1
    assert not zero[:10].any() and not zero[31:].any()
    assert oracle.circle_halfwidths(0).tolist() == [0]


@pytest.mark.parametrize("cx,cy", [(0, 0), (3, 5), (38, 2), (40, 40), (-4, 20), (20, 47)])
def test_circle_clipping_equals_cropped_disc(oracle, cx, cy):
    big = np.ones((141, 141), np.uint8)
    oracle.circle_filled(big, cx + 50, cy + 50, 10)
    small = np.ones((41, 41), np.uint8)
    oracle.circle_filled(small, cx, cy, 10)
    assert np.array_equal(small, big[50:91, 50:91])


@pytest.fixture(scope="module")
def sortidx_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sortidx") / "sortidx_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", out,
                           os.path.join(ROOT, "tests", "cpp", "sortidx_check.cpp")])
    return out


def test_sort_idx_matches_this_toolchains_std_sort(oracle, sortidx_bin):
    rng = np.random.default_rng(4)
    cases = [np.zeros(n, np.int32) for n in list(range(0, 40)) + [63, 64, 65, 100, 255, 256, 300, 511, 599, 600, 601, 1000, 4097]]
    for n in (17, 33, 100, 600, 2000):
        cases.append(rng.integers(0, 3, n).astype(np.int32))          # many ties
        cases.append(rng.integers(-1000, 1000, n).astype(np.int32))   # few ties
        cases.append(np.arange(n, dtype=np.int32))                    # sorted
        cases.append(np.arange(n, dtype=np.int32)[::-1].copy())       # reversed
        cases.append((np.arange(n, dtype=np.int32) % 2))              # organ-pipe like
    text = "".join("%d %s\n" % (len(k), " ".join(map(str, k.tolist()))) for k in cases)
    out = subprocess.run([sortidx_bin], input=text, capture_output=True, text=True, check=True).stdout
    lines = out.split("\n")[:-1]
    assert len(lines) == len(cases)
    for k, line in zip(cases, lines):
        ref = np.array(line.split(), dtype=np.int32) if line.strip() else np.zeros(0, np.int32)
        got = oracle.sort_idx_descending(k)
        assert np.array_equal(got, ref), len(k)
    # equal keys: identity up to 16, a genuine permutation beyond (Q9)
    assert oracle.sort_idx_descending(np.zeros(16, np.int32)).tolist() == list(range(15, -1, -1))
    p = oracle.sort_idx_descending(np.zeros(600, np.int32))
    assert sorted(p.tolist()) == list(range(600)) and p.tolist() != list(range(599, -1, -1))


def _ssc_py(kp, num_ret, tol, cols, rows):
    """anms.cpp:344-438 transcribed with Python floats (doubles) and numpy float32 where C++ has float."""
    exp1 = rows + cols + 2 * num_ret
    exp2 = 4 * cols + 4 * num_ret + 4 * rows * num_ret + rows * rows + cols * cols - 2 * rows * cols + 4 * rows * cols * num_ret
    exp3 = math.sqrt(exp2)
    exp4 = num_ret - 1

    def cround(v):  # C round(): half away from zero
        return math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)
    sol1 = -cround((exp1 + exp3) / exp4)
    sol2 = -cround((exp1 - exp3) / exp4)
    high = int(sol1 if sol1 > sol2 else sol2)
    low = int(math.floor(math.sqrt(len(kp) / num_ret)))
    prev = -1
    K = np.float32(num_ret)
    kmin = int(cround(float(K - K * np.float32(tol))))
    kmax = int(cround(float(K + K * np.float32(tol))))
    result, final = [], []
    while True:
        width = low + int((high - low) / 2)  # C int division truncates toward zero
        if width == prev or low > high:
            final = result
            break
        result = []
        c = width / 2.0
        ncc, ncr = int(math.floor(cols / c)), int(math.floor(rows / c))
        cov = np.zeros((ncr + 1, ncc + 1), bool)
        reach = int(math.floor(width / c))
        for i, (x, y) in enumerate(kp):
            row, col = int(math.floor(float(y) / c)), int(math.floor(float(x) / c))
            if not cov[row, col]:
                result.append(i)
                cov[max(row - reach, 0):min(row + reach, ncr) + 1, max(col - reach, 0):min(col + reach, ncc) + 1] = True
        if kmin <= len(result) <= kmax:
            final = result
            break
        elif len(result) < kmin:
            high = width - 1
        else:
            low = width + 1
        prev = width
    return final


@pytest.mark.parametrize("seed,n,k,cols,rows", [(0, 600, 300, 752, 480), (1, 600, 50, 752, 480), (2, 100, 99, 200, 100),
                                                (3, 40, 7, 64, 48), (4, 600, 299, 1280, 720), (5, 3, 2, 20, 20),
                                                (6, 500, 499, 100, 100)])
def test_ssc_against_python_transcription(oracle, seed, n, k, cols, rows):
    rng = np.random.default_rng(seed)
    pts = set()
    while len(pts) < n:
        pts.add((int(rng.integers(0, cols)), int(rng.integers(0, rows))))
    kp = np.array(sorted(pts), np.float32)
    rng.shuffle(kp)
    got, cnt = oracle.ssc(kp, k, 0.1, cols, rows)
    ref = _ssc_py(kp, k, 0.1, cols, rows)
    assert cnt == len(ref) and got.tolist() == ref


def test_ssc_toy_and_zero_request(oracle):
    # three points, two wanted: the far pair survives at the first width that yields [Kmin, Kmax] = [2, 2]
    kp = np.array([[2, 2], [3, 2], [17, 17]], np.float32)
    got, cnt = oracle.ssc(kp, 2, 0.1, 20, 20)
    assert cnt == 2 and got.tolist() == [0, 2]
    _, cnt = oracle.ssc(kp, 0, 0.1, 20, 20)
    assert cnt == -1  # the reference divides by zero there (undefined); the oracle refuses


def test_anms_mask_rules(oracle):
    det = oracle.AnmsDetector(300, 10.0)
    w, h = 120, 90
    xy = np.array([[60.7, 45.2], [0.0, 10.0], [5.5, 5.5], [119.9, 89.9], [130.0, 40.0]])
    flag = det.build_mask(w, h, xy)
    assert set(np.unique(flag)) <= {0, 1}
    assert not flag[:20].any() and not flag[-20:].any() and not flag[:, :20].any() and not flag[:, -20:].any()
    assert flag[20, 20] == 1 and flag[h - 21, w - 21] == 1
    hw = oracle.circle_halfwidths(10)
    # disc at (int(60.7), int(45.2)) = (60, 45)
    for dy in range(-10, 11):
        for dx in range(-12, 13):
            inside = abs(dx) <= hw[abs(dy)]
            assert flag[45 + dy, 60 + dx] == (0 if inside else 1)
    # user mask: bitwise AND with the edge mask of ones keeps only bit 0 (FeatureDetectorANMS.cpp:132)
    m = np.full((h, w), 255, np.uint8)
    m[30:40, 30:40] = 2
    m[50:60, 30:40] = 0
    m[70:, :] = 3
    flag2 = det.build_mask(w, h, np.zeros((0, 2)), m)
    assert not flag2[30:40, 30:40].any() and not flag2[50:60, 30:40].any()
    assert flag2[25, 25] == 1 and flag2[h - 21, 50] == 1 and flag2[20:30, 20:100].all()


def test_anms_detector_semantics(oracle, synth):
    frames, _ = synth.sequence(3, 320, 240, 2)
    det = oracle.AnmsDetector(100, 10.0)
    T = oracle.FeatureTable(400)
    assert det.update_detection(T, frames[0]) == 0
    n1 = T.n.value
    assert 90 <= n1 <= 110                      # SSC tolerance 0.1 around the 100 requested
    assert T.ids[:n1].tolist() == list(range(1, n1 + 1))  # first id ever issued is 1 (Q6)
    assert det.max_feature_index == n1 and (T.age[:n1] == 1).all()
    xy = T.xy[:n1]
    assert (xy == np.round(xy)).all()
    assert xy[:, 0].min() >= 20 and xy[:, 0].max() <= 320 - 21 and xy[:, 1].min() >= 20 and xy[:, 1].max() <= 240 - 21
    # second call on a thinned table: new corners avoid the discs of the existing ones
    keep = 40
    T.n.value = keep
    old = T.xy[:keep].copy()
    assert det.update_detection(T, frames[1]) == 0
    n2 = T.n.value
    new = T.xy[keep:n2]
    assert n2 > keep and T.ids[keep:n2].tolist() == list(range(n1 + 1, n1 + 1 + n2 - keep))
    flag = det.build_mask(320, 240, old)
    assert all(flag[int(y), int(x)] == 1 for x, y in new)
    # full table: the reference would call Ssc with numRetPoints = 0 (undefined); the oracle adds nothing
    T2 = oracle.FeatureTable(400)
    det2 = oracle.AnmsDetector(5, 10.0)
    T2.set(np.arange(1, 6, dtype=np.uint64), np.full((5, 2), 100.0), np.ones(5, np.uint32))
    assert det2.update_detection(T2, frames[0]) == 1 and T2.n.value == 5 and det2.max_feature_index == 0
    # fewer detections than needed: all of them are taken, in GFTT order, without SSC
    det3 = oracle.AnmsDetector(300, 10.0, detection_feature_number=12)
    T3 = oracle.FeatureTable(400)
    assert det3.update_detection(T3, frames[0]) == 0 and T3.n.value == 12
    flag0 = det3.build_mask(320, 240, np.zeros((0, 2)))
    ref = oracle.good_features_to_track(frames[0], 12, 0.001, 1.0, mask=flag0, block_size=4)
    assert np.array_equal(T3.xy[:12], ref.astype(np.float64))
