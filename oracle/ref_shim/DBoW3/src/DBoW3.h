// ORACLE — TEST INFRASTRUCTURE ONLY.  DBoW3 / OpenCV types appear only as data members of ldso::Frame (loop closing): opaque stand-ins.
#pragma once
#include <map>
#include <vector>
typedef unsigned char uchar;
#define CV_8UC3 16
namespace cv { struct Mat { uchar *data = nullptr; Mat() {} Mat(int, int, int) {} }; }
namespace DBoW3 {
class Vocabulary {};
class BowVector : public std::map<unsigned, double> {};
class FeatureVector : public std::map<unsigned, std::vector<unsigned>> {};
}
