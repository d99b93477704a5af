// TESTS ONLY.  A minimal command-line parser with the names of the TCLAP classes examples/stereo-match/main.cpp:66-133 uses
// (CmdLine, ValueArg<T>, UnlabeledValueArg<T>, SwitchArg, ArgException), so that tests/test_reference_example_compiles.py can compile
// that file -- read from /root/reference at test time -- in an image that does not have TCLAP.  It is not TCLAP, it is not part of the
// product, and nothing under libstevi_amd/ includes it.  Behaviour covered: "--name=value" / "--name value" / "-f value" for ValueArg
// (the delimiter given to CmdLine), bare words for UnlabeledValueArg in order of add(), "--name" / "-f" for SwitchArg, ArgException
// for an unknown flag, a missing value or a missing required argument.
#pragma once

#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace TCLAP {

class ArgException : public std::exception {
  public:
    ArgException(std::string text, std::string id) : _text(std::move(text)), _id(std::move(id)) {}
    std::string error() const { return _text; }
    std::string argId() const { return _id; }
    const char *what() const noexcept override { return _text.c_str(); }

  private:
    std::string _text, _id;
};

class Arg {
  public:
    Arg(std::string flag, std::string name, bool required, bool takesValue, bool positional)
        : _flag(std::move(flag)), _name(std::move(name)), _required(required), _takesValue(takesValue), _positional(positional) {}
    virtual ~Arg() = default;
    virtual void assign(std::string const &text) = 0;
    bool matches(std::string const &token) const {
        return !_positional && ((!_name.empty() && token == "--" + _name) || (!_flag.empty() && token == "-" + _flag));
    }
    std::string id() const { return _positional ? _name : "--" + _name; }
    bool isSet() const { return _set; }

  protected:
    friend class CmdLine;
    std::string _flag, _name;
    bool _required, _takesValue, _positional, _set = false;
};

template <class T> class ValueArg : public Arg {
  public:
    ValueArg(std::string const &flag, std::string const &name, std::string const & /*desc*/, bool req, T value, std::string const & /*typeDesc*/)
        : Arg(flag, name, req, true, false), _value(std::move(value)) {}
    T &getValue() { return _value; }
    void assign(std::string const &text) override {
        if constexpr (std::is_same_v<T, std::string>) {
            _value = text;
        } else {
            std::istringstream in(text);
            T v{};
            if (!(in >> v) || !in.eof()) throw ArgException("Couldn't read argument value from string '" + text + "'", id());
            _value = v;
        }
        _set = true;
    }

  protected:
    ValueArg(std::string const &name, bool req, T value) : Arg("", name, req, true, true), _value(std::move(value)) {}
    T _value;
};

template <class T> class UnlabeledValueArg : public ValueArg<T> {
  public:
    UnlabeledValueArg(std::string const &name, std::string const & /*desc*/, bool req, T value, std::string const & /*typeDesc*/)
        : ValueArg<T>(name, req, std::move(value)) {}
};

class SwitchArg : public Arg {
  public:
    SwitchArg(std::string const &flag, std::string const &name, std::string const & /*desc*/, bool def = false)
        : Arg(flag, name, false, false, false), _value(def), _default(def) {}
    bool getValue() const { return _value; }
    void assign(std::string const &) override {
        _value = !_default;
        _set = true;
    }

  private:
    bool _value, _default;
};

class CmdLine {
  public:
    CmdLine(std::string message, char delimiter = ' ', std::string version = "none") : _message(std::move(message)), _version(std::move(version)), _delimiter(delimiter) {}
    void add(Arg &a) { _args.push_back(&a); }
    void parse(int argc, const char *const *argv) {
        std::size_t nextPositional = 0;
        for (int k = 1; k < argc; k++) {
            std::string token = argv[k], inlineValue;
            bool hasInline = false;
            if (token.size() > 1 && token[0] == '-') {
                const std::size_t cut = token.find(_delimiter);
                if (_delimiter != ' ' && cut != std::string::npos) {
                    inlineValue = token.substr(cut + 1);
                    token = token.substr(0, cut);
                    hasInline = true;
                }
                Arg *hit = nullptr;
                for (Arg *a : _args)
                    if (a->matches(token)) hit = a;
                if (!hit) throw ArgException("Couldn't find match for argument", token);
                if (!hit->_takesValue) {
                    hit->assign("");
                } else if (hasInline) {
                    hit->assign(inlineValue);
                } else {
                    if (k + 1 >= argc) throw ArgException("Missing a value for this argument!", hit->id());
                    hit->assign(argv[++k]);
                }
                continue;
            }
            while (nextPositional < _args.size() && !_args[nextPositional]->_positional) nextPositional++;
            if (nextPositional >= _args.size()) throw ArgException("Too many unlabeled arguments", token);
            _args[nextPositional++]->assign(token);
        }
        for (Arg *a : _args)
            if (a->_required && !a->isSet()) throw ArgException("Required argument missing", a->id());
    }

  private:
    std::string _message, _version;
    char _delimiter;
    std::vector<Arg *> _args;
};

} // namespace TCLAP
