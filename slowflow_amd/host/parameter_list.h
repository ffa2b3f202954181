// parameter_list.h -- the cfg surface of the drop-in: a from-scratch ParameterList with the reference's file
// syntax and accessors (utils/parameter_list.h:20-143, parameter_list.cpp:34-229, 658-723):
//   key<TAB>value[<TAB># comment]     one per line; '#' lines are comments; "(a,b,c)" = experiment grid values;
//   file / output / start / Jets / F / center / extent / verbose are also kept as named members.
// Values are kept as strings and parsed on every access, exactly like the reference (the solver writes
// parameters back: "final", "slow_flow_img_norm_*").  No OpenCV: Point2f / Point are two-field structs.
#ifndef SLOWFLOW_AMD_HOST_PARAMETER_LIST_H
#define SLOWFLOW_AMD_HOST_PARAMETER_LIST_H

#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct Point2f { float x, y; Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {} };
struct Point { int x, y; Point(int x_ = 0, int y_ = 0) : x(x_), y(y_) {} };
inline std::ostream &operator<<(std::ostream &os, const Point2f &p) { return os << "[" << p.x << ", " << p.y << "]"; }

enum Verbosity { VER_CMD = 0, VER_IN_GT = 1, VER_IMG_PYR = 2, VER_FLO_PYR = 3, WRITE_FILES = 4 };

class ParameterList {
public:
    ParameterList();
    explicit ParameterList(const std::string &file);

    void read(const std::string &file);

    void insert(const std::string &param, const std::string &val, bool overwrite = false);
    void insert(const std::string &param, const std::vector<std::string> &vals, bool overwrite = false);
    template <typename T> void setParameter(const std::string &param, T value) {
        std::stringstream v;
        v << value;
        setParameterString(param, v.str());
    }
    bool exists(const std::string &param) const;
    bool verbosity(unsigned state) const { return state < verbose.size() && verbose[state] == '1'; }

    std::string parameter(const char *param) const;                        // "" + error message when missing
    template <typename T> T parameter(const std::string &param) const { return parameter<T>(param, ""); }
    template <typename T> T parameter(const std::string &param, const std::string &def) const;
    template <typename T> std::vector<T> splitParameter(const std::string &param, const std::string &def = "") const;

    // experiment grid ("(a,b,c)" values): same iteration interface as the reference
    unsigned experiments() const { return (unsigned)exps; }
    unsigned experiment() const { return (unsigned)current_exp; }
    bool hasNextExp() const { return current_exp + 1 < exps; }
    bool nextExp();
    void reset();

    void print() const;
    std::string cfgString() const;
    friend std::ostream &operator<<(std::ostream &os, const ParameterList &p) { return os << p.cfgString(); }

    // named parameters (utils/parameter_list.h:102-128)
    std::string verbose;
    std::string file;
    std::vector<std::string> file_list;
    unsigned sequence_start = 0;
    std::vector<unsigned> sequence_start_list;
    std::string output;
    unsigned F = 0;
    unsigned Jets = 0;
    std::string file_gt;
    Point center, extent;

private:
    void setParameterString(const std::string &param, const std::string &value);
    const std::string *current(const std::string &param) const;
    static std::vector<std::string> parse(const std::string &value);

    std::vector<std::string> order;                              // insertion order
    std::map<std::string, std::vector<std::string>> values;      // all values of a key
    std::map<std::string, unsigned> selected;                    // which one is current
    int exps = 1, current_exp = 0;
};

#endif
