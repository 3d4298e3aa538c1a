// ORACLE — TEST INFRASTRUCTURE ONLY.  DBoW3 / OpenCV types appear only as data members of ldso::Frame (loop closing): opaque stand-ins.
#pragma once
#include <map>
#include <vector>
#include <fstream>
#include <unistd.h>
typedef unsigned char uchar;
#define CV_8UC3 16
#define CV_8U 0
namespace cv { struct Mat { uchar *data = nullptr; Mat() {} Mat(int, int, int) {} };
// cv::RNG appears in the random point-selection branch of makeNewTraces (FullSystem.cc:1308-1312, setting_pointSelection == 2): never taken in the pin
struct RNG { unsigned long long st = 0xffffffffull; int uniform(int a, int b) { st = st * 4164903690ull + (st >> 32); return a + (int) ((unsigned) st % (unsigned) (b - a)); } }; }
namespace DBoW3 {
class BowVector; class FeatureVector;
class Vocabulary { public: template <class D> void transform(const D &, BowVector &, FeatureVector &, int) const {} };     // Frame::ComputeBoW (loop closing only)
class Database {};        // LoopClosing.h:104 (member type only)
typedef unsigned int EntryId;
class BowVector : public std::map<unsigned, double> {};
class FeatureVector : public std::map<unsigned, std::vector<unsigned>> {};
}
