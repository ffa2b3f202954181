// parameter_list.cpp -- see parameter_list.h
#include "parameter_list.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

using std::string;
using std::vector;

ParameterList::ParameterList() : verbose(string(10, '0')), center(-1, -1), extent(-1, -1) {}
ParameterList::ParameterList(const string &filename) : ParameterList() { read(filename); }

static vector<string> split_tabs(const string &line) {
    vector<string> out;
    size_t pos = 0;
    while (pos <= line.size()) {
        size_t next = line.find('\t', pos);
        if (next == string::npos) next = line.size();
        if (next > pos) out.push_back(line.substr(pos, next - pos));      // consecutive tabs collapse, like strtok
        pos = next + 1;
    }
    return out;
}

void ParameterList::read(const string &filename) {
    std::ifstream f(filename.c_str(), std::ios::binary);
    if (!f) { perror("Error opening file"); return; }
    string line;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        const vector<string> tok = split_tabs(line);
        if (tok.empty() || tok[0].empty() || tok[0][0] == '#') continue;
        const string &name = tok[0];
        if (tok.size() < 2 || tok[1][0] == '#') {
            std::cerr << "Value to parameter '" << name << "' is missing!" << std::endl;
            continue;
        }
        const string &value = tok[1];
        // named keys (parameter_list.cpp:53-208)
        if (name == "file") { file = value; file_list.push_back(value); continue; }
        if (name == "file_gt") { file_gt = value; continue; }
        if (name == "output") { output = value; continue; }
        if (name == "start") { sequence_start = (unsigned)atoi(value.c_str()); sequence_start_list.push_back(sequence_start); continue; }
        if (name == "F") { F = (unsigned)atoi(value.c_str()); continue; }
        if (name == "Jets") { Jets = (unsigned)atoi(value.c_str()); continue; }
        if (name == "center" || name == "extent") {
            const size_t comma = value.find(',');
            if (comma != string::npos) {
                Point p((int)atof(value.substr(0, comma).c_str()), (int)atof(value.substr(comma + 1).c_str()));
                (name == "center" ? center : extent) = p;
            }
            continue;
        }
        if (name == "verbose") verbose = value;                   // and kept as a generic parameter too
        insert(name, parse(value), true);
    }
}

vector<string> ParameterList::parse(const string &value) {
    vector<string> out;
    if (!value.empty() && value[0] == '(') {                      // "(a,b,c)": a list of experiment values
        string inner = value.substr(1);
        const size_t close = inner.find(')');
        if (close != string::npos) inner = inner.substr(0, close);
        size_t pos = 0;
        while (pos <= inner.size()) {
            size_t next = inner.find(',', pos);
            if (next == string::npos) next = inner.size();
            if (next > pos) out.push_back(inner.substr(pos, next - pos));
            pos = next + 1;
        }
    } else out.push_back(value);
    return out;
}

void ParameterList::insert(const string &param, const string &val, bool overwrite) { insert(param, vector<string>(1, val), overwrite); }

void ParameterList::insert(const string &param, const vector<string> &vals, bool overwrite) {
    auto it = values.find(param);
    if (it == values.end()) {
        order.push_back(param);
        values[param] = vals;
        selected[param] = 0;
        exps *= (int)vals.size();
    } else {
        exps /= (int)it->second.size();
        if (overwrite) { it->second = vals; selected[param] = 0; }
        else it->second.insert(it->second.end(), vals.begin(), vals.end());
        exps *= (int)it->second.size();
    }
}

bool ParameterList::exists(const string &param) const { return values.find(param) != values.end(); }

const string *ParameterList::current(const string &param) const {
    auto it = values.find(param);
    if (it == values.end() || it->second.empty()) return nullptr;
    return &it->second[selected.at(param)];
}

void ParameterList::setParameterString(const string &param, const string &value) {
    auto it = values.find(param);
    if (it == values.end()) { insert(param, value, true); return; }
    it->second[selected[param]] = value;
}

string ParameterList::parameter(const char *param) const {
    const string *v = current(param);
    if (!v) { std::cerr << "Error: Parameter " << param << " does not exist!" << std::endl; return ""; }
    return *v;
}

// typed getters: missing key -> default string if given, else an error message and 0 (parameter_list.cpp:669-723)
template <> string ParameterList::parameter<string>(const string &param, const string &def) const {
    const string *v = current(param);
    return v ? *v : def;
}
template <> int ParameterList::parameter<int>(const string &param, const string &def) const {
    const string *v = current(param);
    if (v) return atoi(v->c_str());
    if (!def.empty()) return atoi(def.c_str());
    std::cerr << "Error: Parameter " << param << " does not exist!" << std::endl;
    return 0;
}
template <> double ParameterList::parameter<double>(const string &param, const string &def) const {
    const string *v = current(param);
    if (v) return atof(v->c_str());
    if (!def.empty()) return atof(def.c_str());
    std::cerr << "Error: Parameter " << param << " does not exist!" << std::endl;
    return 0;
}
template <> float ParameterList::parameter<float>(const string &param, const string &def) const {
    const string *v = current(param);
    if (v) return (float)atof(v->c_str());
    if (!def.empty()) return (float)atof(def.c_str());
    std::cerr << "Error: Parameter " << param << " does not exist!" << std::endl;
    return 0;
}
template <> bool ParameterList::parameter<bool>(const string &param, const string &def) const {
    const string *v = current(param);
    if (v) return *v != "0";
    if (!def.empty()) return def != "0";
    std::cerr << "Error: Parameter " << param << " does not exist!" << std::endl;
    return false;
}
template <> vector<int> ParameterList::splitParameter<int>(const string &param, const string &def) const {
    const string *v = current(param);
    const string s = v ? *v : def;
    vector<int> out;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t next = s.find(',', pos);
        if (next == string::npos) next = s.size();
        if (next > pos) out.push_back(atoi(s.substr(pos, next - pos).c_str()));
        pos = next + 1;
    }
    return out;
}
template <> vector<float> ParameterList::splitParameter<float>(const string &param, const string &def) const {
    const string *v = current(param);
    const string s = v ? *v : def;
    vector<float> out;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t next = s.find(',', pos);
        if (next == string::npos) next = s.size();
        if (next > pos) out.push_back((float)atof(s.substr(pos, next - pos).c_str()));
        pos = next + 1;
    }
    return out;
}

bool ParameterList::nextExp() {
    if (!hasNextExp()) return false;
    current_exp++;
    for (const string &k : order) {                               // odometer over the multi-valued keys
        auto &vals = values[k];
        if (vals.size() < 2) continue;
        if (++selected[k] < vals.size()) return true;
        selected[k] = 0;
    }
    return true;
}
void ParameterList::reset() {
    current_exp = 0;
    for (auto &s : selected) s.second = 0;
}

string ParameterList::cfgString() const {
    std::stringstream os;
    os << "file\t" << file << "\n" << "output\t" << output << "\n" << "start\t" << sequence_start << "\n" << "Jets\t" << Jets << "\n";
    if (center.x >= 0) os << "center\t" << center.x << "," << center.y << "\n";
    if (extent.x >= 0) os << "extent\t" << extent.x << "," << extent.y << "\n";
    for (const string &k : order) {
        const auto &vals = values.at(k);
        os << k << "\t";
        if (vals.size() == 1) os << vals[0];
        else {
            os << "(";
            for (size_t i = 0; i < vals.size(); i++) os << (i ? "," : "") << vals[i];
            os << ")";
        }
        os << "\n";
    }
    return os.str();
}
void ParameterList::print() const { std::cout << cfgString() << std::endl; }
