// cli_support.h -- console helpers of the render CLI: ANSI colours, the
// "<job>... took N ms." phase lines and the key/value device table.
//
// The reference spreads these over include/{color,timer,info}.h; only what is
// observable matters here: the phase names and line shapes of
// reference src/info.cc:14-42 (they delimit what "Rendering image" times) and
// the millisecond wall clock of reference src/timer.cc:11-18.
//
// Everything lives in namespace ocrt::cli: the reference defines global `Color`, `Info`
// and `Timer` classes of its own (src/color.cc, info.cc, timer.cc, with a different
// layout), and a maintainer who links those files next to libocrt_hip.so must not get
// two definitions of one symbol.
#pragma once
#include <cstddef>
#include <functional>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace ocrt {
namespace cli {

struct Color {
	static const char *RESET, *RED, *GREEN, *YELLOW, *BLUE, *PURPLE, *CYAN, *WHITE;
};

// Wall-clock stopwatch, whole milliseconds.
class Timer {
	public:
		Timer() { reset(); }
		void reset() { start = now(); }
		static std::size_t now();
		std::size_t get_elapsed() const { return now() - start; }
	private:
		std::size_t start;
};

class Info {
	public:
		// Runs job, prints "<description>... took N ms." and returns N.  With
		// synchronous = true the job's own output is bracketed by "(1/2)"/"(2/2)"
		// lines.  A job returning false prints " failed!" and exits the process.
		static std::size_t measure(const std::string &description, const std::function<bool()> &job,
		                           bool synchronous = false);
		static std::string formatTime(std::size_t elapsed_ms);

		void setTitle(const std::string &t) { title = t; }
		template <class T> void add(const std::string &name, const T &value) {
			std::stringstream ss;
			ss << value;
			attributes.emplace_back(name, ss.str());
		}
		void add(const Info &child) { children.push_back(child); }
		std::string str() const;

		struct Palette {
			static const char *NORMAL, *HIGHLIGHT, *SECTION, *WARNING;
		};
	private:
		std::size_t longestName() const;
		std::string title;
		std::vector<std::pair<std::string, std::string>> attributes;
		std::vector<Info> children;
};

}  // namespace cli
}  // namespace ocrt
