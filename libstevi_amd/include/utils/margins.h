// StereoVision::Margins / PaddingMargins -- value types that appear in the hot path's signatures
// (reference: utils/margins.h:24-163).  Same constructors and accessors, written for the drop-in headers.
#pragma once

namespace StereoVision {

class Margins {
  public:
    Margins() {}
    Margins(int all) : l(all), t(all), r(all), b(all) {}
    Margins(int leftright, int topbottom) : l(leftright), t(topbottom), r(leftright), b(topbottom) {}
    Margins(int left, int top, int right, int bottom) : l(left), t(top), r(right), b(bottom) {}
    int left() const { return l; }
    int top() const { return t; }
    int right() const { return r; }
    int bottom() const { return b; }

  protected:
    int l = 0, t = 0, r = 0, b = 0;
};

// default-constructed = "automatic" padding (the window radii), anything else is explicit
class PaddingMargins : public Margins {
  public:
    PaddingMargins() : Margins(), automatic(true) {}
    PaddingMargins(int all) : Margins(all), automatic(false) {}
    PaddingMargins(int leftright, int topbottom) : Margins(leftright, topbottom), automatic(false) {}
    PaddingMargins(int left, int top, int right, int bottom) : Margins(left, top, right, bottom), automatic(false) {}
    PaddingMargins(Margins const &m) : Margins(m), automatic(false) {}
    bool isAuto() const { return automatic; }

  private:
    bool automatic;
};

} // namespace StereoVision
