#include "cli_support.h"

#include <sys/time.h>

#include <algorithm>
#include <cstdlib>
#include <iostream>

namespace ocrt {
namespace cli {

#ifndef NO_COLORS
const char *Color::RESET = "\033[0m";
const char *Color::RED = "\033[0;91m";
const char *Color::GREEN = "\033[0;92m";
const char *Color::YELLOW = "\033[0;93m";
const char *Color::BLUE = "\033[0;94m";
const char *Color::PURPLE = "\033[0;95m";
const char *Color::CYAN = "\033[0;96m";
const char *Color::WHITE = "\033[0;97m";
#else
const char *Color::RESET = "";
const char *Color::RED = "";
const char *Color::GREEN = "";
const char *Color::YELLOW = "";
const char *Color::BLUE = "";
const char *Color::PURPLE = "";
const char *Color::CYAN = "";
const char *Color::WHITE = "";
#endif
const char *Info::Palette::NORMAL = Color::WHITE;
const char *Info::Palette::HIGHLIGHT = Color::YELLOW;
const char *Info::Palette::SECTION = Color::GREEN;
const char *Info::Palette::WARNING = Color::RED;

std::size_t Timer::now() {
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	return (std::size_t) ((tv.tv_sec * 1000000ull + (unsigned long long) tv.tv_usec) / 1000ull);
}

std::size_t Info::measure(const std::string &description, const std::function<bool()> &job, bool synchronous) {
	const std::string status = std::string(Palette::NORMAL) + description + "\xE2\x80\xA6" + Color::RESET;
	if (synchronous)
		std::cout << status << " (1/2)" << std::endl;
	else
		std::cout << status;
	Timer timer;
	const bool ok = job();
	const std::size_t elapsed = timer.get_elapsed();
	if (synchronous)
		std::cout << status;
	if (!ok)
		std::cout << Palette::WARNING << " failed!" << Color::RESET;
	std::cout << (synchronous ? " (2/2)" : "") << " took " << formatTime(elapsed) << "." << std::endl;
	if (!ok)
		std::exit(EXIT_FAILURE);
	return elapsed;
}

std::string Info::formatTime(std::size_t elapsed_ms) {
	std::stringstream ss;
	ss << Color::GREEN << elapsed_ms << " ms" << Color::RESET;
	return ss.str();
}

std::size_t Info::longestName() const {
	std::size_t longest = 0;
	for (const auto &a : attributes)
		longest = std::max(longest, a.first.size());
	for (const Info &c : children)
		longest = std::max(longest, c.longestName());
	return longest;
}

std::string Info::str() const {
	const std::size_t column = longestName() + 3;
	std::stringstream ss;
	ss << Color::YELLOW << "*** " << Color::GREEN << title << Color::YELLOW << " ***" << Color::RESET << std::endl;
	ss << Color::BLUE << std::string(title.size() + 8, '=') << Color::RESET << std::endl;
	for (const auto &a : attributes) {
		std::string value = a.second;
		if (value.size() > 100)
			value = value.substr(0, 99) + "\xE2\x80\xA6";
		ss << Color::WHITE << a.first << Color::RESET << std::string(column - a.first.size(), '.') << Color::GREEN
		   << value << Color::RESET << std::endl;
	}
	if (!children.empty())
		ss << std::endl;
	for (const Info &c : children)
		ss << c.str() << std::endl;
	return ss.str();
}

}  // namespace cli
}  // namespace ocrt
