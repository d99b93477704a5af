// Drop-in for LibStevi's utils/contiguity.h: which neighbours of a pixel count as contiguous (Rook = the 4 edge neighbours,
// Bishop = the 4 corner neighbours, Queen = all 8), the direction tables and their sizes.  Host-only constants; the reference's
// correlation/cross_correlations.h includes this header (cross_correlations.h:29) and examples/stereo-match/main.cpp:192 names
// StereoVision::Contiguity::Queen through it.
//
// Same names and values as the reference (utils/contiguity.h:26-195): enum generalContiguity, enum bidimensionalContiguity,
// BidimensionalContiguityTraits<c>::{nDir, nCornerDir, nTilingDir}, nDirections / nCornerDirections / nTilingDirections,
// getDirections<c>() / getCornerDirections<c>() / getTilingDirections<c>() with the rows in the reference's order.  Built here from
// one table of the eight neighbour offsets filtered by a predicate instead of one explicit specialisation per table.
#ifndef STEREOVISION_UTILS_CONTIGUITY_H
#define STEREOVISION_UTILS_CONTIGUITY_H

#include <array>

namespace StereoVision {

class Contiguity {
  public:
    enum generalContiguity { singleDimCanChange, allDimsCanChange };

    enum bidimensionalContiguity { Rook, Bishop, Queen };

    constexpr static int nDirections(bidimensionalContiguity contiguity) { return contiguity == Queen ? 8 : 4; }
    constexpr static int nCornerDirections(bidimensionalContiguity contiguity) { return contiguity == Queen ? 3 : (contiguity == Rook ? 2 : 1); }
    constexpr static int nTilingDirections(bidimensionalContiguity contiguity) { return contiguity == Queen ? 4 : 2; }

    template <bidimensionalContiguity contiguity> class BidimensionalContiguityTraits {
      public:
        static constexpr int nDir = nDirections(contiguity);
        static constexpr int nCornerDir = nCornerDirections(contiguity);
        static constexpr int nTilingDir = nTilingDirections(contiguity);
    };

    // all neighbours, rows from (+1, +1) down to (-1, -1) -- the order of the reference's Queen table, of which the Rook and Bishop
    // tables are the sub-sequences with one / two non-zero offsets (utils/contiguity.h:97-136)
    template <bidimensionalContiguity contiguity> constexpr static std::array<std::array<int, 2>, nDirections(contiguity)> getDirections() {
        return select<nDirections(contiguity)>(contiguity, allNeighbours, 8);
    }

    // the directions towards the lower-right corner: (1,1), (1,0), (0,1) filtered (utils/contiguity.h:140-163)
    template <bidimensionalContiguity contiguity> constexpr static std::array<std::array<int, 2>, nCornerDirections(contiguity)> getCornerDirections() {
        constexpr int corner[3][2] = {{1, 1}, {1, 0}, {0, 1}};
        return select<nCornerDirections(contiguity)>(contiguity, corner, 3);
    }

    // one direction per undirected neighbour pair: (1,1), (1,0), (0,1), (1,-1) filtered (utils/contiguity.h:167-190)
    template <bidimensionalContiguity contiguity> constexpr static std::array<std::array<int, 2>, nTilingDirections(contiguity)> getTilingDirections() {
        constexpr int tiling[4][2] = {{1, 1}, {1, 0}, {0, 1}, {1, -1}};
        return select<nTilingDirections(contiguity)>(contiguity, tiling, 4);
    }

  private:
    static constexpr int allNeighbours[8][2] = {{1, 1}, {1, 0}, {1, -1}, {0, 1}, {0, -1}, {-1, 1}, {-1, 0}, {-1, -1}};

    constexpr static bool belongs(bidimensionalContiguity contiguity, int di, int dj) {
        const bool diagonal = di != 0 && dj != 0;
        return contiguity == Queen || (contiguity == Bishop ? diagonal : !diagonal);
    }
    template <int nOut> constexpr static std::array<std::array<int, 2>, nOut> select(bidimensionalContiguity contiguity, const int (*table)[2], int n) {
        std::array<std::array<int, 2>, nOut> out{};
        int k = 0;
        for (int t = 0; t < n; t++) {
            if (!belongs(contiguity, table[t][0], table[t][1])) continue;
            if (k < nOut) out[k] = {table[t][0], table[t][1]};
            k++;
        }
        return out;
    }
};

} // namespace StereoVision

#endif // STEREOVISION_UTILS_CONTIGUITY_H
