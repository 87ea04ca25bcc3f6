// Test helper (CPU): prints what cv::sortIdx(keys, idx, CV_SORT_DESCENDING) does with THIS toolchain's
// libstdc++: idx = iota; std::sort(idx, LessThanIdx(keys)); reverse (OpenCV 3.4 core/src/matrix_sort
// sortIdx_, one row).  tests/test_oracle_anms.py compares oracle/vsf_oracle.c's restatement of
// std::sort against it.  Input: n, then n keys.  Output: n indices.
#include <algorithm>
#include <cstdio>
#include <vector>
int main() {
  int n;
  while (std::scanf("%d", &n) == 1) {
    std::vector<int> keys(n), idx(n);
    for (int i = 0; i < n; i++) {
      if (std::scanf("%d", &keys[i]) != 1) return 1;
      idx[i] = i;
    }
    const int* arr = keys.data();
    std::sort(idx.begin(), idx.end(), [arr](int a, int b) { return arr[a] < arr[b]; });
    for (int j = 0; j < n / 2; j++) std::swap(idx[j], idx[n - 1 - j]);
    for (int i = 0; i < n; i++) std::printf("%d ", idx[i]);
    std::printf("\n");
  }
  return 0;
}
